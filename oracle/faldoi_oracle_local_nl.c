/*
 * faldoi_oracle_local_nl.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of the local step's other functionals: of_estimation's branches for -m 2..7
 * (src/energy_model.cpp:691-802) and the tables prepare_stuff builds for them (:276-659):
 *
 *   M_NLTVL1 (2)      guided_nltvl1      src/nltv_model.cpp:275       lambda 2,    theta 0.3, tau 0.1
 *   M_NLTVL1_W (3)    guided_nltvl1_w    src/nltvw_model.cpp          lambda 0.85 / c^2 (the `lambda = 2; lambda = 0.85;` of :771-772)
 *   M_TVCSAD (4)      guided_tvcsad      src/tvcsad_model.cpp:265     lambda 0.85
 *   M_TVCSAD_W (5)    guided_tvcsad_w    src/tvcsadw_model.cpp        lambda 0.85 / c^2
 *   M_NLTVCSAD (6)    guided_nltvcsad    src/nltvcsad_model.cpp:297   lambda 0.85
 *   M_NLTVCSAD_W (7)  guided_nltvcsad_w  src/nltvcsadw_model.cpp      lambda 0.85 / c^2
 *   (c = weight[w_radio + 1] with the radius in force during the call, like -m 1)
 *
 * Which libm overload each call binds to is part of the arithmetic.  The model files and aux_energy_model.cpp include
 * <cmath> only and have no using-directive, so their unqualified sqrt / exp / fabs / hypot / pow on float arguments are
 * the C double functions; energy_model.cpp additionally includes mask.h -> <math.h> (inside extern "C"), which pulls
 * the float overloads into the global namespace, so rgb_to_lab's exp is expf while its pow(x, 1./3) stays double.
 * Every function below is pinned bit for bit against the reference's own compiled objects (oracle/_ref, tests/
 * test_oracle_models_cpu.py).
 *
 * State a caller keeps per growing direction: for the non-local functionals the dual values p.sc / q.sc (24 + 24
 * planes; prepare_stuff zeroes the valid slots, later calls continue from what earlier windows left) and the constant
 * weights wp (24 planes); TV-CSAD keeps nothing that matters -- its duals are zeroed per call and, the reference never
 * computing the flow gradients its dual step reads (u1x..u2y are the zeros of a fresh allocation), they
 * stay zero, so its regulariser is inert and both divergence planes are zero.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "faldoi_oracle.h"

#define NL_R 2            /* NL_BETA, src/parameters.h:82 */
#define NL_SLOTS 24       /* NL_DUAL_VAR */
#define CS_R 3            /* DT_R */
#define CS_SLOTS 48       /* DT_NEI */
#define GRAD_ZERO 1E-8    /* GRAD_IS_ZERO */

/* rgb_to_lab, src/energy_model.cpp:60-90 (planar in, planar out) */
void orc_rgb_to_lab_local(const float *in, int size, float *out)
{
    const float T = 0.008856;
    const float color_attenuation = 1.5f;
    for (int i = 0; i < size; i++) {
        const float r = in[i] / 255.f;
        const float g = in[i + size] / 255.f;
        const float b = in[i + 2 * size] / 255.f;
        float X = 0.412453 * r + 0.357580 * g + 0.180423 * b;
        float Y = 0.212671 * r + 0.715160 * g + 0.072169 * b;
        float Z = 0.019334 * r + 0.119193 * g + 0.950227 * b;
        X /= 0.950456;
        Z /= 1.088754;
        float Y3 = pow(Y, 1. / 3);
        float fX = X > T ? pow(X, 1. / 3) : 7.787 * X + 16 / 116.;
        float fY = Y > T ? Y3 : 7.787 * Y + 16 / 116.;
        float fZ = Z > T ? pow(Z, 1. / 3) : 7.787 * Z + 16 / 116.;
        float L = Y > T ? 116 * Y3 - 16.0 : 903.3 * Y;
        float A = 500 * (fX - fY);
        float B = 200 * (fY - fZ);
        const float l2 = (L / 100) * (L / 100);                 /* pow2(L / 100): float */
        const float d = l2 - 0.6;                               /* double difference, passed to pow2(float) */
        const float correct_lab = expf(-color_attenuation * (d * d));   /* exp(float) is the float overload in that file */
        out[i] = L;
        out[i + size] = A * correct_lab;
        out[i + 2 * size] = B * correct_lab;
    }
}

/* nltv_ini_dual_variables, src/aux_energy_model.cpp:112-175 with get_weight :91, get_wcolor :63, get_wspatial :49.
 * wp[slot][n] (0 where the neighbour is outside the image; the reference leaves -2 there and never reads it),
 * wt[n] = sum over the valid slots.  Slot order: rows k = -2..2, columns l = -2..2, centre skipped. */
void orc_nltv_table_local(const float *a, int pd, int w, int h, float *wp, float *wt)
{
    const size_t n = (size_t)w * h;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            const size_t pos = (size_t)j * w + i;
            int it = 0;
            float tot = 0.0;
            for (int k = -NL_R; k <= NL_R; k++)
                for (int l = -NL_R; l <= NL_R; l++) {
                    if (k == 0 && l == 0) continue;
                    float v = 0.0f;
                    if (i + l >= 0 && i + l < w && j + k >= 0 && j + k < h) {
                        float difI = 0.0;
                        for (int m = 0; m < pd; m++) {
                            const float aux = a[n * m + pos] - a[n * m + (size_t)(j + k) * w + (i + l)];
                            difI += aux * aux;
                        }
                        difI = sqrt(difI);                              /* double sqrt, stored as float */
                        const float wc = exp(-difI / 2.0f);             /* NL_INTENSITY = 2; double exp of a float quotient */
                        float difS = hypot(l, k);
                        const float ws = exp(-difS / 2.0f);             /* NL_BETA = 2 */
                        v = sqrt(wc * ws);                              /* float product, double sqrt */
                        tot += v;
                    }
                    wp[(size_t)it * n + pos] = v;
                    it++;
                }
            wt[pos] = tot;
        }
}

/* ---- the 48-slot CSAD step on a window: b terms, then per iteration the selection of sorted entry n + 1 ---- */
static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* One of_estimation call.  method 2..7.  duals: [48][nx*ny] (p slots then q slots) for the non-local functionals, else
 * NULL; wp: [24][nx*ny] or NULL.  weight / iiw / ijw / wr_call: as orc_local_tvl1w_estimate (used by 3, 5, 7).
 * u1, u2 are updated inside the window; returns the window energy. */
float orc_local_nl_estimate(int method, const float *I0, const float *I1, const float *I1x, const float *I1y, float *u1,
                            float *u2, float *duals, const float *wp, int nx, int ny, int ii, int ij, int ei, int ej,
                            int max_iter_patch, const float *weight, int iiw, int ijw, int wr_call, int *iters_out)
{
    const int nl = (method == 2 || method == 3 || method == 6 || method == 7);
    const int csad = (method >= 4);
    const int l1 = !csad;
    const int weighted = (method & 1);
    const int sqrt_form = (method == 7);            /* plain |grad|^2 with sqrt() divisions and the GRAD_IS_ZERO guard */
    const size_t n = (size_t)nx * ny;
    float lambda = (method == 2) ? 2 : 0.85;
    const float theta = 0.3, tau = 0.1, tol_OF = 0.01;
    if (weighted) {
        const float central = weight[wr_call + 1];
        lambda = lambda / (central * central);
    }
    const float l_t = lambda * theta;
    const int pw = ei - ii, ph = ej - ij, np = pw * ph;
    float *I1w = malloc(sizeof(float) * np * 16), *I1wx = I1w + np, *I1wy = I1wx + np, *grad = I1wy + np, *rho_c = grad + np;
    float *v1 = rho_c + np, *v2 = v1 + np, *ub1 = v2 + np, *ub2 = ub1 + np, *t1 = ub2 + np, *t2 = t1 + np, *dp = t2 + np, *dq = dp + np;
    float *wt = dq + np;
    float *b = csad ? malloc(sizeof(float) * np * CS_SLOTS) : NULL;
    int *cnt = malloc(sizeof(int) * np);
    float ba[2 * CS_SLOTS + 1];
    float *ps = duals, *qs = duals ? duals + (size_t)NL_SLOTS * n : NULL;
#define WIN(l, k) (((l) - ij) * pw + ((k) - ii))
#define INWIN(k, l) ((k) >= ii && (k) < ei && (l) >= ij && (l) < ej)

    for (int l = ij; l < ej; l++)
        for (int k = ii; k < ei; k++) {
            const size_t i = (size_t)l * nx + k;
            const int q = WIN(l, k);
            const float uu = (float)k + u1[i], vv = (float)l + u2[i];
            I1w[q] = orc_bicubic_at(I1, uu, vv, nx, ny, 0);
            I1wx[q] = orc_bicubic_at(I1x, uu, vv, nx, ny, 0);
            I1wy[q] = orc_bicubic_at(I1y, uu, vv, nx, ny, 0);
        }
    for (int l = ij; l < ej; l++)
        for (int k = ii; k < ei; k++) {
            const size_t i = (size_t)l * nx + k;
            const int q = WIN(l, k);
            const float Ix2 = I1wx[q] * I1wx[q];
            const float Iy2 = I1wy[q] * I1wy[q];
            if (l1) {
                grad[q] = (Ix2 + Iy2);
                rho_c[q] = (I1w[q] - I1wx[q] * u1[i] - I1wy[q] * u2[i] - I0[i]);
                continue;
            }
            grad[q] = sqrt_form ? (Ix2 + Iy2) : (float)hypot(Ix2 + Iy2, 0.01);
            cnt[q] = 0;
            if (sqrt_form && !(grad[q] > GRAD_ZERO)) continue;
            int s = 0, m = 0;
            for (int kk = -CS_R; kk <= CS_R; kk++)
                for (int ll = -CS_R; ll <= CS_R; ll++) {
                    if (kk == 0 && ll == 0) continue;
                    if (INWIN(k + ll, l + kk)) {
                        const size_t pos = (size_t)(l + kk) * nx + (k + ll);
                        const int qq = WIN(l + kk, k + ll);
                        const float num = I0[i] - I0[pos] - I1w[q] + I1w[qq] + I1wx[q] * u1[i] + I1wy[q] * u2[i];
                        b[(size_t)q * CS_SLOTS + s] = sqrt_form ? (float)(num / sqrt(grad[q])) : num / grad[q];
                        m++;
                    }
                    s++;
                }
            cnt[q] = m;
        }
    if (nl)                                          /* wt = sum of the weights of the in-window neighbours */
        for (int l = ij; l < ej; l++)
            for (int k = ii; k < ei; k++) {
                const size_t i = (size_t)l * nx + k;
                float acc = 0.0;
                int s = 0;
                for (int kk = -NL_R; kk <= NL_R; kk++)
                    for (int ll = -NL_R; ll <= NL_R; ll++) {
                        if (kk == 0 && ll == 0) continue;
                        if (INWIN(k + ll, l + kk)) acc += wp[(size_t)s * n + i];
                        s++;
                    }
                wt[WIN(l, k)] = acc;
            }
    for (int l = ij; l < ej; l++)
        for (int k = ii; k < ei; k++) {
            const size_t i = (size_t)l * nx + k;
            ub1[WIN(l, k)] = u1[i];
            ub2[WIN(l, k)] = u2[i];
        }

    int it_n = 0;
    float err_D = INFINITY;
    while (err_D > tol_OF * tol_OF && it_n < max_iter_patch) {
        it_n++;
        for (int l = ij; l < ej; l++)
            for (int k = ii; k < ei; k++) {
                const size_t i = (size_t)l * nx + k;
                const int q = WIN(l, k);
                const float l_t_w = weighted ? l_t * weight[l - ij + ijw] * weight[k - ii + iiw] : l_t;
                if (l1) {
                    const float rho = rho_c[q] + (I1wx[q] * u1[i] + I1wy[q] * u2[i]);
                    float d1, d2;
                    if (rho < -l_t_w * grad[q]) {
                        d1 = l_t_w * I1wx[q];
                        d2 = l_t_w * I1wy[q];
                    } else if (rho > l_t_w * grad[q]) {
                        d1 = -l_t_w * I1wx[q];
                        d2 = -l_t_w * I1wy[q];
                    } else if (grad[q] < GRAD_ZERO) {
                        d1 = d2 = 0;
                    } else {
                        const float fi = -rho / grad[q];
                        d1 = fi * I1wx[q];
                        d2 = fi * I1wy[q];
                    }
                    v1[q] = u1[i] + d1;
                    v2[q] = u2[i] + d2;
                    continue;
                }
                v1[q] = u1[i];
                v2[q] = u2[i];
                if (sqrt_form && !(grad[q] > GRAD_ZERO)) continue;
                int cntq = 0, s = 0;
                const float dot = I1wx[q] * u1[i] + I1wy[q] * u2[i];
                for (int kk = -CS_R; kk <= CS_R; kk++)
                    for (int ll = -CS_R; ll <= CS_R; ll++) {
                        if (kk == 0 && ll == 0) continue;
                        if (INWIN(k + ll, l + kk)) {
                            const float bj = b[(size_t)q * CS_SLOTS + s];
                            ba[cntq++] = sqrt_form ? (float)(-(bj - dot / sqrt(grad[q]))) : -(bj - dot / grad[q]);
                        }
                        s++;
                    }
                const int m = cnt[q];
                for (int j = 0; j < m + 1; j++)
                    ba[cntq++] = sqrt_form ? (float)((m - 2 * j) * l_t_w * sqrt(grad[q])) : (m - 2 * j) * l_t_w * grad[q];
                qsort(ba, cntq, sizeof(float), cmp_float);
                const float med = ba[cntq / 2 + 1];
                if (sqrt_form) {
                    v1[q] = u1[i] - I1wx[q] * med / sqrt(grad[q]);
                    v2[q] = u2[i] - I1wy[q] * med / sqrt(grad[q]);
                } else {
                    v1[q] = u1[i] - I1wx[q] * med / grad[q];
                    v2[q] = u2[i] - I1wy[q] * med / grad[q];
                }
            }
        if (nl)                                      /* *_getD on ubar, in-window neighbours only */
            for (int l = ij; l < ej; l++)
                for (int k = ii; k < ei; k++) {
                    const size_t i = (size_t)l * nx + k;
                    const int q = WIN(l, k);
                    int s = 0;
                    for (int kk = -NL_R; kk <= NL_R; kk++)
                        for (int ll = -NL_R; ll <= NL_R; ll++) {
                            if (kk == 0 && ll == 0) continue;
                            if (INWIN(k + ll, l + kk)) {
                                const int qq = WIN(l + kk, k + ll);
                                const float w_ = wp[(size_t)s * n + i];
                                const float nlgr1 = w_ * (ub1[q] - ub1[qq]) / wt[q];
                                const float nlgr2 = w_ * (ub2[q] - ub2[qq]) / wt[q];
                                const float nl1 = sqrt(nlgr1 * nlgr1);
                                const float nl2 = sqrt(nlgr2 * nlgr2);
                                const float nl1g = 1 + tau * nl1;
                                const float nl2g = 1 + tau * nl2;
                                ps[(size_t)s * n + i] = (ps[(size_t)s * n + i] + tau * nlgr1) / nl1g;
                                qs[(size_t)s * n + i] = (qs[(size_t)s * n + i] + tau * nlgr2) / nl2g;
                            }
                            s++;
                        }
                }
        for (int l = ij; l < ej; l++)
            for (int k = ii; k < ei; k++) {
                const size_t i = (size_t)l * nx + k;
                t1[WIN(l, k)] = u1[i];
                t2[WIN(l, k)] = u2[i];
            }
        if (nl)                                      /* non_local_divergence, src/aux_energy_model.cpp:178 (no division by wt) */
            for (int l = ij; l < ej; l++)
                for (int k = ii; k < ei; k++) {
                    const size_t i = (size_t)l * nx + k;
                    float a1 = 0.0, a2 = 0.0;
                    int s = 0;
                    for (int kk = -NL_R; kk <= NL_R; kk++)
                        for (int ll = -NL_R; ll <= NL_R; ll++) {
                            if (kk == 0 && ll == 0) continue;
                            if (INWIN(k + ll, l + kk)) {
                                const size_t pos = (size_t)(l + kk) * nx + (k + ll);
                                const float w_ = wp[(size_t)s * n + i];
                                const int rp = NL_SLOTS - (s + 1);
                                a1 += w_ * (ps[(size_t)s * n + i] - ps[(size_t)rp * n + pos]);
                                a2 += w_ * (qs[(size_t)s * n + i] - qs[(size_t)rp * n + pos]);
                            }
                            s++;
                        }
                    dp[WIN(l, k)] = a1;
                    dq[WIN(l, k)] = a2;
                }
        err_D = 0.0;
        for (int l = ij; l < ej; l++)
            for (int k = ii; k < ei; k++) {
                const size_t i = (size_t)l * nx + k;
                const int q = WIN(l, k);
                const float u1k = u1[i], u2k = u2[i];
                if (nl) {
                    u1[i] = u1k - tau * (dp[q] + (u1k - v1[q]) / theta);
                    u2[i] = u2k - tau * (dq[q] + (u2k - v2[q]) / theta);
                } else {                             /* TV-CSAD: -div of the inert regulariser is -0 */
                    u1[i] = u1k - tau * (-0.0f + (u1k - v1[q]) / theta);
                    u2[i] = u2k - tau * (-0.0f + (u2k - v2[q]) / theta);
                }
                err_D += (u1[i] - u1k) * (u1[i] - u1k) + (u2[i] - u2k) * (u2[i] - u2k);
            }
        err_D /= (ej - ij) * (ei - ii);
        for (int l = ij; l < ej; l++)
            for (int k = ii; k < ei; k++) {
                const size_t i = (size_t)l * nx + k;
                const int q = WIN(l, k);
                ub1[q] = 2 * u1[i] - t1[q];
                ub2[q] = 2 * u2[i] - t2[q];
            }
    }
    if (iters_out) *iters_out = it_n;

    /* eval_*: warp once more by the new flow, mean window energy */
    for (int l = ij; l < ej; l++)
        for (int k = ii; k < ei; k++) {
            const size_t i = (size_t)l * nx + k;
            I1w[WIN(l, k)] = orc_bicubic_at(I1, (float)k + u1[i], (float)l + u2[i], nx, ny, 0);
        }
    float ener = 0.0;
    int m = 0;
    for (int l = ij; l < ej; l++)
        for (int k = ii; k < ei; k++) {
            const size_t i = (size_t)l * nx + k;
            const int q = WIN(l, k);
            const float dc = (1 / (2 * theta)) * ((u1[i] - v1[q]) * (u1[i] - v1[q]) + (u2[i] - v2[q]) * (u2[i] - v2[q]));
            float g = 0.0;
            if (nl) {
                int s = 0;
                for (int kk = -NL_R; kk <= NL_R; kk++)
                    for (int ll = -NL_R; ll <= NL_R; ll++) {
                        if (kk == 0 && ll == 0) continue;
                        if (INWIN(k + ll, l + kk)) {
                            const size_t pos = (size_t)(l + kk) * nx + (k + ll);
                            const float w_ = wp[(size_t)s * n + i];
                            g += fabs(u1[i] - u1[pos]) * w_ + fabs(u2[i] - u2[pos]) * w_;      /* double products and sum */
                        }
                        s++;
                    }
                g /= wt[q];
            }
            float dt;
            if (l1) {
                dt = weighted ? lambda * fabs(I1w[q] - I0[i]) * weight[l - ij + ijw] * weight[k - ii + iiw]
                              : lambda * fabs(I1w[q] - I0[i]);
            } else {
                dt = 0.0;
                for (int kk = -CS_R; kk <= CS_R; kk++)
                    for (int ll = -CS_R; ll <= CS_R; ll++) {
                        if (kk == 0 && ll == 0) continue;
                        if (INWIN(k + ll, l + kk)) {
                            const size_t pos = (size_t)(l + kk) * nx + (k + ll);
                            dt += fabs(I0[i] - I0[pos] - I1w[q] + I1w[WIN(l + kk, k + ll)]);
                        }
                    }
                if (weighted) dt *= lambda * weight[l - ij + ijw] * weight[k - ii + iiw];
                else dt *= lambda;
            }
            ener += dc + dt + g;
            m++;
        }
    ener /= (m * 1.0);
    free(I1w);
    free(b);
    free(cnt);
    return ener;
#undef WIN
#undef INWIN
}
