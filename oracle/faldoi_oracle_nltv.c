/*
 * faldoi_oracle_nltv.c -- CPU restatement of the NLTV-CSAD global solver (BASELINE configs[4]).
 *
 * TEST INFRASTRUCTURE ONLY (see faldoi_oracle.h).  Restates, with SoA planes and implicit
 * neighbour geometry instead of the reference's AoS index tables:
 *   image_to_lab                 src/global_faldoi.cpp:906-932
 *   weights of the 5x5 duals     src/global_faldoi.cpp:935-1054 (get_wspatial_2, get_wcolor_2,
 *                                initialize_dual_variables)
 *   7x7 CSAD neighbour table     src/global_faldoi.cpp:1332-1374 (initialize_pos_nei)
 *   ofnltv_getD / non_local_divergence / ofnltv_getP   :1127-1174 / :1056-1079 / :1090-1120
 *   nltvcsad_PD                  src/global_faldoi.cpp:1642-1808
 *
 * Slot numbering (both tables): offsets enumerated rows k=-r..r, columns l=-r..r, centre skipped,
 * the slot counter advances for EVERY non-centre offset, valid or not (:1026-1047), so slot <->
 * offset is a fixed table and the mirrored slot of s is n_slots-1-s (rp, :1033).
 *
 * libm calls are the ones the C++ reference resolves to: pow(double), expf (exp of a float
 * argument), hypot(double), sqrtf.  Compile with -ffp-contract=off.
 */
#include "faldoi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NL_R 2            /* MAX_BETA, :887 */
#define NL_SLOTS 24       /* MAX_DUAL_VAR, :888 */
#define CS_R 3            /* DT_R, src/parameters.h:86 */
#define CS_SLOTS 48       /* DT_NEI, src/parameters.h:87 */

static inline float sq(float f) { return f * f; }     /* aux_pow2, :904 */

/* src/global_faldoi.cpp:906-932.  in: planar R,G,B (0..255); out: planar L, a*, b* */
void orc_image_to_lab(const float *in, int size, float *out)
{
    const float T = 0.008856f;
    const float color_attenuation = 1.5f;
    for (int i = 0; i < size; i++) {
        const float r = in[i] / 255.f;
        const float g = in[i + size] / 255.f;
        const float b = in[i + 2 * size] / 255.f;
        float X = (float)(0.412453 * r + 0.357580 * g + 0.180423 * b);
        float Y = (float)(0.212671 * r + 0.715160 * g + 0.072169 * b);
        float Z = (float)(0.019334 * r + 0.119193 * g + 0.950227 * b);
        X = (float)(X / 0.950456);
        Z = (float)(Z / 1.088754);
        const float Y3 = (float)pow(Y, 1. / 3);
        const float fX = (float)(X > T ? pow(X, 1. / 3) : 7.787 * X + 16 / 116.);
        const float fY = (float)(Y > T ? (double)Y3 : 7.787 * Y + 16 / 116.);
        const float fZ = (float)(Z > T ? pow(Z, 1. / 3) : 7.787 * Z + 16 / 116.);
        const float L = (float)(Y > T ? 116 * Y3 - 16.0 : 903.3 * Y);
        const float A = 500 * (fX - fY);
        const float B = 200 * (fY - fZ);
        /* aux_pow2 takes a float: the double (L/100)^2 - 0.6 is narrowed before squaring */
        const float inner = (float)(sq(L / 100) - 0.6);
        const float correct_lab = expf(-color_attenuation * sq(inner));
        out[i] = L;
        out[i + size] = A * correct_lab;
        out[i + 2 * size] = B * correct_lab;
    }
}

/* Weights of the 24 non-local dual slots and their sum (:935-1054).  wts: [24][n] (0 for slots
 * whose neighbour is outside the image), wt: [n].  `a` = Lab image, pd channels, planar. */
void orc_nltv_weights(const float *a, int pd, int w, int h, float *wts, float *wt)
{
    const int n = w * h;
#pragma omp parallel for
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            const int pos = j * w + i;
            int it = 0;
            float ne = 0.0f;
            for (int k = -NL_R; k <= NL_R; k++)
                for (int l = -NL_R; l <= NL_R; l++) {
                    if (k == 0 && l == 0) continue;
                    float wp = 0.0f;
                    if (i + l >= 0 && i + l < w && j + k >= 0 && j + k < h) {
                        float difI = 0.0f;                                   /* get_wcolor_2 :958-983 */
                        for (int m = 0; m < pd; m++) {
                            const float aux = a[n * m + pos] - a[n * m + (j + k) * w + (i + l)];
                            difI += aux * aux;
                        }
                        difI = sqrtf(difI);
                        const float wc = expf(-difI / 5.0f);                 /* MAX_INTENSITY */
                        const float difS = (float)hypot((double)l, (double)k);  /* get_wspatial_2 :946-956 */
                        const float ws = expf(-difS / 2.0f);                 /* MAX_BETA */
                        wp = sqrtf(wc * ws);                                 /* :1040 */
                        ne += wp;
                    }
                    wts[(size_t)it * n + pos] = wp;
                    it++;
                }
            wt[pos] = ne;
        }
}

static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

static void nl_solver(int data_l1, const float *I0, const float *I1, const float *lab, int pd,
                      float lambda, float theta, float tau, int w, int h, int warps, int max_inner,
                      float *u1, float *u2);

/* src/global_faldoi.cpp:1642-1808.  u1,u2 in/out.  Duals start at zero (:1031). */
void orc_nltvcsad_global(const float *I0, const float *I1, const float *lab, int pd,
                         float lambda, float theta, float tau, int w, int h, int warps, int max_inner,
                         float *u1, float *u2)
{
    nl_solver(0, I0, I1, lab, pd, lambda, theta, tau, w, h, warps, max_inner, u1, u2);
}

/* NLTV-L1, nltvl1_PD, src/global_faldoi.cpp:1177-1327: the same non-local regulariser with the
 * TV-L1 data term -- rho_c (:1232-1233) and the thresholding step (:1246-1272, identical to
 * tvl2OF's :690-718) replace the CSAD selection. */
void orc_nltvl1_global(const float *I0, const float *I1, const float *lab, int pd,
                       float lambda, float theta, float tau, int w, int h, int warps, int max_inner,
                       float *u1, float *u2)
{
    nl_solver(1, I0, I1, lab, pd, lambda, theta, tau, w, h, warps, max_inner, u1, u2);
}

static void nl_solver(int data_l1, const float *I0, const float *I1, const float *lab, int pd,
                      float lambda, float theta, float tau, int w, int h, int warps, int max_inner,
                      float *u1, float *u2)
{
    const int n = w * h;
    const float l_t = lambda * theta;
    float *buf = (float *)calloc((size_t)n * (2 * NL_SLOTS + NL_SLOTS + 1 + CS_SLOTS + 13), sizeof(float));
    float *p = buf, *q = p + (size_t)NL_SLOTS * n, *wts = q + (size_t)NL_SLOTS * n, *wt = wts + (size_t)NL_SLOTS * n;
    float *b = wt + n;
    float *I1x = b + (size_t)CS_SLOTS * n, *I1y = I1x + n, *I1w = I1y + n, *I1wx = I1w + n, *I1wy = I1wx + n;
    float *grad = I1wy + n, *v1 = grad + n, *v2 = v1 + n, *ub1 = v2 + n, *ub2 = ub1 + n, *t1 = ub2 + n, *t2 = t1 + n;
    float *divs = t2 + n;                      /* div_p reused for div_q: 1 plane, computed inline below */
    (void)divs;

    orc_nltv_weights(lab, pd, w, h, wts, wt);
    orc_centered_gradient(I1, I1x, I1y, w, h);                                   /* :1690 */

    for (int wp_ = 0; wp_ < warps; wp_++) {
        orc_bicubic_warp(I1, u1, u2, I1w, w, h, 1);                              /* :1694-1696 */
        orc_bicubic_warp(I1x, u1, u2, I1wx, w, h, 1);
        orc_bicubic_warp(I1y, u1, u2, I1wy, w, h, 1);
#pragma omp parallel for
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int i = y * w + x;
                const float Ix2 = I1wx[i] * I1wx[i], Iy2 = I1wy[i] * I1wy[i];
                grad[i] = Ix2 + Iy2;
                if (data_l1)                                                    /* rho_c reuses the I1w plane */
                    I1w[i] = I1w[i] - I1wx[i] * u1[i] - I1wy[i] * u2[i] - I0[i];
                if (!data_l1 && grad[i] > 1E-8) {                               /* :1704 */
                    int s = 0;
                    for (int k = -CS_R; k <= CS_R; k++)
                        for (int l = -CS_R; l <= CS_R; l++) {
                            if (k == 0 && l == 0) continue;
                            if (x + l >= 0 && x + l < w && y + k >= 0 && y + k < h) {
                                const int pos = (y + k) * w + (x + l);
                                b[(size_t)s * n + i] = (I0[i] - I0[pos] - I1w[i] + I1w[pos] + I1wx[i] * u1[i]
                                                        + I1wy[i] * u2[i]) / sqrtf(grad[i]);   /* :1716-1717 */
                            }
                            s++;
                        }
                }
                ub1[i] = u1[i];                                                  /* :1722-1725 */
                ub2[i] = u2[i];
            }

        for (int it_ = 0; it_ < max_inner; it_++) {                              /* :1729 (no tolerance test) */
            /* v-step :1733-1758 */
#pragma omp parallel for
            for (int y = 0; y < h; y++) {
                float ba[2 * CS_SLOTS + 1];
                for (int x = 0; x < w; x++) {
                    const int i = y * w + x;
                    v1[i] = u1[i];
                    v2[i] = u2[i];
                    if (data_l1) {                                               /* :1246-1272 */
                        const float rho = I1w[i] + (I1wx[i] * u1[i] + I1wy[i] * u2[i]);
                        float d1, d2;
                        if (rho < -l_t * grad[i]) { d1 = l_t * I1wx[i]; d2 = l_t * I1wy[i]; }
                        else if (rho > l_t * grad[i]) { d1 = -l_t * I1wx[i]; d2 = -l_t * I1wy[i]; }
                        else if (grad[i] < 1E-8) { d1 = d2 = 0; }
                        else { const float fi = -rho / grad[i]; d1 = fi * I1wx[i]; d2 = fi * I1wy[i]; }
                        v1[i] = u1[i] + d1;
                        v2[i] = u2[i] + d2;
                    } else if (grad[i] > 1E-8) {
                        const float sg = sqrtf(grad[i]);
                        int it = 0, s = 0, nn = 0;
                        for (int k = -CS_R; k <= CS_R; k++)
                            for (int l = -CS_R; l <= CS_R; l++) {
                                if (k == 0 && l == 0) continue;
                                if (x + l >= 0 && x + l < w && y + k >= 0 && y + k < h) {
                                    ba[it++] = -(b[(size_t)s * n + i] - (I1wx[i] * u1[i] + I1wy[i] * u2[i]) / sg);
                                    nn++;
                                }
                                s++;
                            }
                        for (int j = 0; j < nn + 1; j++) ba[it++] = (nn - 2 * j) * l_t * sg;
                        qsort(ba, (size_t)it, sizeof(float), cmp_float);
                        const float m = ba[it / 2 + 1];                          /* :1755, sic */
                        v1[i] = u1[i] - I1wx[i] * m / sg;
                        v2[i] = u2[i] - I1wy[i] * m / sg;
                    }
                }
            }
            /* ofnltv_getD on ubar :1761 (:1127-1174) */
#pragma omp parallel for
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    const int i = y * w + x;
                    int s = 0;
                    for (int k = -NL_R; k <= NL_R; k++)
                        for (int l = -NL_R; l <= NL_R; l++) {
                            if (k == 0 && l == 0) continue;
                            if (x + l >= 0 && x + l < w && y + k >= 0 && y + k < h) {
                                const int ap = (y + k) * w + (x + l);
                                const float wgt = wts[(size_t)s * n + i];
                                const float g1 = wgt * (ub1[i] - ub1[ap]) / wt[i];
                                const float d1 = 1 + tau * sqrtf(g1 * g1);
                                p[(size_t)s * n + i] = (p[(size_t)s * n + i] + tau * g1) / d1;
                                const float g2 = wgt * (ub2[i] - ub2[ap]) / wt[i];
                                const float d2 = 1 + tau * sqrtf(g2 * g2);
                                q[(size_t)s * n + i] = (q[(size_t)s * n + i] + tau * g2) / d2;
                            }
                            s++;
                        }
                }
            /* :1764-1778: u_tmp = u; divergences (:1056-1079); getP (:1090-1120); relaxation */
#pragma omp parallel for
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    const int i = y * w + x;
                    float dp = 0.0f, dq = 0.0f;
                    int s = 0;
                    for (int k = -NL_R; k <= NL_R; k++)
                        for (int l = -NL_R; l <= NL_R; l++) {
                            if (k == 0 && l == 0) continue;
                            if (x + l >= 0 && x + l < w && y + k >= 0 && y + k < h) {
                                const int ap = (y + k) * w + (x + l);
                                const int rp = NL_SLOTS - 1 - s;
                                const float wgt = wts[(size_t)s * n + i];
                                dp += wgt * (p[(size_t)s * n + i] - p[(size_t)rp * n + ap]);
                                dq += wgt * (q[(size_t)s * n + i] - q[(size_t)rp * n + ap]);
                            }
                            s++;
                        }
                    dp /= wt[i];
                    dq /= wt[i];
                    const float u1k = u1[i], u2k = u2[i];
                    t1[i] = u1k - tau * (dp + (u1k - v1[i]) / theta);
                    t2[i] = u2k - tau * (dq + (u2k - v2[i]) / theta);
                }
#pragma omp parallel for
            for (int i = 0; i < n; i++) {
                ub1[i] = 2 * t1[i] - u1[i];
                ub2[i] = 2 * t2[i] - u2[i];
                u1[i] = t1[i];
                u2[i] = t2[i];
            }
        }
    }
    free(buf);
}


/* ------------------------------------------------------------------------------------------
 * TV-CSAD, tvcsad_PD, src/global_faldoi.cpp:1449-1636 (global_faldoi -m 4 / 5): the CSAD data term of
 * the NLTV-CSAD solver above over sg = hypot(Ix^2 + Iy^2, 0.01) (double hypot, :1512) with a TV
 * regulariser whose duals are normalised per flow component with hypotf (:1424-1441), and the stop test
 * mean |du|^2 > tol^2 (:1540, :1386-1414).
 * The reference accumulates that mean with `err_D +=` inside an `omp parallel for` without a reduction
 * clause, i.e. racily; it is deterministic only on one thread, where the terms are added in pixel
 * order.  That order is what is restated here (the sum runs serially; the rest may use threads).
 * xi: 4 planes in/out (main zeroes them, :2116-2121); they persist over the warps.
 * ---------------------------------------------------------------------------------------- */
long orc_tvcsad_global(const float *I0, const float *I1, float lambda, float theta, float tau, float tol_OF,
                       int w, int h, int warps, int max_inner, float *u1, float *u2, float *xi,
                       int *iters_out, float *err_out)
{
    const int n = w * h;
    const float l_t = lambda * theta;
    float *buf = (float *)calloc((size_t)n * (CS_SLOTS + 18), sizeof(float));
    float *b = buf;
    float *I1x = b + (size_t)CS_SLOTS * n, *I1y = I1x + n, *I1w = I1y + n, *I1wx = I1w + n, *I1wy = I1wx + n;
    float *grad = I1wy + n, *v1 = grad + n, *v2 = v1 + n, *ub1 = v2 + n, *ub2 = ub1 + n;
    float *u1x = ub2 + n, *u1y = u1x + n, *u2x = u1y + n, *u2y = u2x + n, *dv1 = u2y + n, *dv2 = dv1 + n, *tt = dv2 + n;
    float *xi11 = xi, *xi12 = xi + n, *xi21 = xi + 2 * (size_t)n, *xi22 = xi + 3 * (size_t)n;
    long total = 0;

    orc_centered_gradient(I1, I1x, I1y, w, h);                                   /* :1504 */
    for (int wp_ = 0; wp_ < warps; wp_++) {
        orc_bicubic_warp(I1, u1, u2, I1w, w, h, 1);                              /* :1509-1511 */
        orc_bicubic_warp(I1x, u1, u2, I1wx, w, h, 1);
        orc_bicubic_warp(I1y, u1, u2, I1wy, w, h, 1);
#pragma omp parallel for
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int i = y * w + x;
                const float Ix2 = I1wx[i] * I1wx[i], Iy2 = I1wy[i] * I1wy[i];
                grad[i] = (float)hypot(Ix2 + Iy2, 0.01);                         /* :1518 */
                int s = 0;
                for (int k = -CS_R; k <= CS_R; k++)
                    for (int l = -CS_R; l <= CS_R; l++) {
                        if (k == 0 && l == 0) continue;
                        if (x + l >= 0 && x + l < w && y + k >= 0 && y + k < h) {
                            const int pos = (y + k) * w + (x + l);
                            b[(size_t)s * n + i] = (I0[i] - I0[pos] - I1w[i] + I1w[pos] + I1wx[i] * u1[i]
                                                    + I1wy[i] * u2[i]) / grad[i];   /* :1529-1530 */
                        }
                        s++;
                    }
                ub1[i] = u1[i];                                                  /* :1535-1538 */
                ub2[i] = u2[i];
            }
        int it_n = 0;
        float err_D = INFINITY;
        while (err_D > tol_OF * tol_OF && it_n < max_inner) {                    /* :1542 */
            it_n++;
#pragma omp parallel for
            for (int y = 0; y < h; y++) {                                        /* :1547-1571 */
                float ba[2 * CS_SLOTS + 1];
                for (int x = 0; x < w; x++) {
                    const int i = y * w + x;
                    int it = 0, s = 0, nn = 0;
                    for (int k = -CS_R; k <= CS_R; k++)
                        for (int l = -CS_R; l <= CS_R; l++) {
                            if (k == 0 && l == 0) continue;
                            if (x + l >= 0 && x + l < w && y + k >= 0 && y + k < h) {
                                ba[it++] = -(b[(size_t)s * n + i] - (I1wx[i] * u1[i] + I1wy[i] * u2[i]) / grad[i]);
                                nn++;
                            }
                            s++;
                        }
                    for (int j = 0; j < nn + 1; j++) ba[it++] = (nn - 2 * j) * l_t * grad[i];
                    qsort(ba, (size_t)it, sizeof(float), cmp_float);
                    v1[i] = u1[i] - I1wx[i] * ba[it / 2 + 1] / grad[i];          /* :1568-1569, sic */
                    v2[i] = u2[i] - I1wy[i] * ba[it / 2 + 1] / grad[i];
                }
            }
            orc_forward_gradient(ub1, u1x, u1y, w, h);                           /* :1576-1578 */
            orc_forward_gradient(ub2, u2x, u2y, w, h);
#pragma omp parallel for
            for (int i = 0; i < n; i++) {                                        /* tvcsad_getD :1424-1441 */
                float n1 = hypotf(xi11[i], xi12[i]), n2 = hypotf(xi21[i], xi22[i]);
                n1 = (1 > n1) ? 1 : n1;
                n2 = (1 > n2) ? 1 : n2;
                xi11[i] = (xi11[i] + tau * u1x[i]) / n1;
                xi12[i] = (xi12[i] + tau * u1y[i]) / n1;
                xi21[i] = (xi21[i] + tau * u2x[i]) / n2;
                xi22[i] = (xi22[i] + tau * u2y[i]) / n2;
            }
            orc_divergence(xi11, xi12, dv1, w, h);                               /* :1581-1582 */
            orc_divergence(xi21, xi22, dv2, w, h);
#pragma omp parallel for
            for (int i = 0; i < n; i++) {                                        /* tvcsad_getP :1386-1414 + :1593-1597 */
                const float u1k = u1[i], u2k = u2[i];
                u1[i] = u1k - tau * (-dv1[i] + (u1k - v1[i]) / theta);
                u2[i] = u2k - tau * (-dv2[i] + (u2k - v2[i]) / theta);
                tt[i] = (u1[i] - u1k) * (u1[i] - u1k) + (u2[i] - u2k) * (u2[i] - u2k);
                ub1[i] = 2 * u1[i] - u1k;
                ub2[i] = 2 * u2[i] - u2k;
            }
            err_D = 0.0f;
            for (int i = 0; i < n; i++) err_D += tt[i];                          /* one thread: pixel order */
            err_D /= n;
        }
        if (iters_out) iters_out[wp_] = it_n;
        if (err_out) err_out[wp_] = err_D;
        total += it_n;
    }
    free(buf);
    return total;
}
