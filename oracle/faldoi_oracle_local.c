/*
 * faldoi_oracle_local.c -- CPU restatement of the LOCAL TV-L1 patch operator:
 * of_estimation's default branch (src/energy_model.cpp:796-800) = guided_tvl2coupled
 * (src/tvl2_model.cpp:249-435) followed by eval_tvl2coupled (:174-243), on one patch window
 * [ii,ei) x [ij,ej) of full-size planes, with the patch stencils forward_gradient_patch
 * (src/utils.cpp:175-220) and divergence_patch (src/utils.cpp:63-113).
 *
 * TEST INFRASTRUCTURE ONLY (see faldoi_oracle.h).
 *
 * The operator is stateful in the reference: its scratch planes (Tvl2CoupledOFStuff) live as long
 * as the growing runs, and divergence_patch does not write the patch's own first row and first /
 * last column (it writes the IMAGE's row 0, column 0 and column nx-1 instead), so
 * those entries of div keep whatever an earlier call left there.  The state is therefore explicit
 * here: `st` = 6 persistent planes (xi11, xi12, xi21, xi22, div1, div2), zero at creation.
 */
#include "faldoi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/utils.cpp:175-220: zero fx on the PATCH's last column, fy on its last row */
static void fwd_grad_patch(const float *f, float *fx, float *fy, int ii, int ij, int ei, int ej, int nx)
{
    for (int j = ij; j < ej; j++)
        for (int i = ii; i < ei; i++) {
            const int p = j * nx + i;
            fx[p] = (i < ei - 1) ? f[p + 1] - f[p] : 0.0f;
            fy[p] = (j < ej - 1) ? f[p + nx] - f[p] : 0.0f;
        }
}

/* src/utils.cpp:63-113, statement by statement (order matters where targets coincide) */
static void div_patch(const float *v1, const float *v2, float *div, int ii, int ij, int ei, int ej, int nx)
{
    for (int j = ij + 1; j < ej - 1; j++)                                   /* :76-86 body */
        for (int i = ii + 1; i < ei - 1; i++) {
            const int p = j * nx + i;
            div[p] = (v1[p] - v1[p - 1]) + (v2[p] - v2[p - nx]);
        }
    for (int i = ii + 1; i < ei - 1; i++) {                                 /* :88-93: IMAGE row 0, patch last row */
        const int p = (ej - 1) * nx + i;
        div[i] = v1[i] - v1[i - 1] + v2[i];
        div[p] = v1[p] - v1[p - 1] - v2[p - nx];
    }
    for (int j = ij + 1; j < ej - 1; j++) {                                 /* :96-103: IMAGE columns 0 and nx-1 */
        const int p1 = j * nx;
        const int p2 = (j + 1) * nx - 1;
        div[p1] = v1[p1] + v2[p1] - v2[p1 - nx];
        div[p2] = -v1[p2 - 1] + v2[p2] - v2[p2 - nx];
    }
    div[ij * nx + ii] = v1[ij * nx + ii] + v2[ij * nx + ii];               /* :104-108 the four patch corners */
    div[ij * nx + ei - 1] = -v1[ij * nx + ei - 2] + v2[ij * nx + ei - 1];
    div[(ej - 1) * nx + ii] = v1[(ej - 1) * nx + ii] - v2[(ej - 2) * nx + ii];
    div[(ej - 1) * nx + ei - 1] = -v1[(ej - 1) * nx + ei - 2] - v2[(ej - 2) * nx + ei - 1];
}

/* One call of of_estimation (TV-l2 coupled).  I1x/I1y = centred gradient of I1 (prepare_stuff).
 * u1,u2: full planes, updated inside the window.  st: 6*nx*ny persistent floats.  Returns the
 * patch energy (*ener_N of the reference); *iters_out = inner iterations run. */
static float local_estimate(const float *I0, const float *I1, const float *I1x, const float *I1y,
                            float *u1, float *u2, float *st, int nx, int ny,
                            int ii, int ij, int ei, int ej, int max_iter_patch, int *iters_out,
                            const float *weight, int iiw, int ijw, float lambda);

float orc_local_tvl1_estimate(const float *I0, const float *I1, const float *I1x, const float *I1y,
                              float *u1, float *u2, float *st, int nx, int ny,
                              int ii, int ij, int ei, int ej, int max_iter_patch, int *iters_out)
{
    return local_estimate(I0, I1, I1x, I1y, u1, u2, st, nx, ny, ii, ij, ei, ej, max_iter_patch, iters_out, NULL, 0, 0, 40);
}

/* gaussian1Dweight as prepare_stuff calls it (src/energy_model.cpp:432-435): src/mask.c:360-372, the C-linkage copy --
 * energy_model.cpp includes mask.h inside extern "C" before utils.h (:18-23), so the name binds to the C function, where
 * exp() is the double function (the C++ copy, src/utils.cpp:633, would call expf: some weights differ in the last bit).
 * -(i-r)*(i-r) is an int, the quotient by `den` a float, exp and the product run in double, one rounding to float. */
void orc_gaussian1d_weight(float *I, int r)
{
    const float sigma = r * 0.3333;
    const float den = 2 * sigma * sigma;
    for (int i = 0; i < 2 * r + 1; i++)
        I[i] = (float)(1 / (sigma * sqrt(2.0 * 3.1415926)) * exp((double)(-(i - r) * (i - r) / den)));
}

/* The weighted functional (-m 1): of_estimation's M_TVL1_W branch (src/energy_model.cpp:742-749) =
 * guided_tvl2coupled_w (src/tvl2w_model.cpp:254-455) + eval_tvl2coupled_w (:173-250).  Same iteration as above
 * with l_t scaled per pixel by weight[row]*weight[column] (:379) and the data energy weighted (:227);
 * weight: gaussian1Dweight of the configured radius; iiw/ijw: get_relative_index_weight
 * (src/local_faldoi.cpp:551-557) for the radius of this call; wr_call: that radius (the reference divides lambda
 * by weight[w_radio + 1]^2 with the w_radio in force during the call, :744-745). */
float orc_local_tvl1w_estimate(const float *I0, const float *I1, const float *I1x, const float *I1y,
                               float *u1, float *u2, float *st, int nx, int ny,
                               int ii, int ij, int ei, int ej, int max_iter_patch, int *iters_out,
                               const float *weight, int iiw, int ijw, int wr_call)
{
    const float central = weight[wr_call + 1];
    const float lambda = 40.0f / (central * central);
    return local_estimate(I0, I1, I1x, I1y, u1, u2, st, nx, ny, ii, ij, ei, ej, max_iter_patch, iters_out, weight, iiw, ijw,
                          lambda);
}

static float local_estimate(const float *I0, const float *I1, const float *I1x, const float *I1y,
                            float *u1, float *u2, float *st, int nx, int ny,
                            int ii, int ij, int ei, int ej, int max_iter_patch, int *iters_out,
                            const float *weight, int iiw, int ijw, float lambda)
{
    const float theta = 0.3f, tau = 0.125f, tol_OF = 0.01f;                /* src/energy_model.cpp:704-709 */
    const float l_t = lambda * theta;
    const size_t n = (size_t)nx * ny;
    float *xi11 = st, *xi12 = st + n, *xi21 = st + 2 * n, *xi22 = st + 3 * n, *div1 = st + 4 * n, *div2 = st + 5 * n;
    /* scratch planes: kept per thread between calls (the local growing makes millions of calls); every
     * entry that is read is written first in the same call, except v when no iteration runs */
    static __thread float *w = NULL;
    static __thread size_t w_n = 0;
    if (w_n != n) {
        free(w);
        w = (float *)calloc(n * 13, sizeof(float));
        w_n = n;
    }
    float *I1w = w, *I1wx = w + n, *I1wy = w + 2 * n, *grad = w + 3 * n, *rho_c = w + 4 * n, *ub1 = w + 5 * n,
          *ub2 = w + 6 * n, *v1 = w + 7 * n, *v2 = w + 8 * n, *u1x = w + 9 * n, *u1y = w + 10 * n, *u2x = w + 11 * n,
          *u2y = w + 12 * n;

    for (int l = ij; l < ej; l++)                                          /* :313-321 */
        for (int k = ii; k < ei; k++) {
            const int i = l * nx + k;
            xi11[i] = xi12[i] = xi21[i] = xi22[i] = 0.0f;
            if (max_iter_patch <= 0) v1[i] = v2[i] = 0.0f;
        }
    for (int l = ij; l < ej; l++)                                          /* :325-355, warps = 1 */
        for (int k = ii; k < ei; k++) {
            const int i = l * nx + k;
            const float uu = (float)k + u1[i], vv = (float)l + u2[i];
            I1w[i] = orc_bicubic_at(I1, uu, vv, nx, ny, 0);
            I1wx[i] = orc_bicubic_at(I1x, uu, vv, nx, ny, 0);
            I1wy[i] = orc_bicubic_at(I1y, uu, vv, nx, ny, 0);
            const float Ix2 = I1wx[i] * I1wx[i], Iy2 = I1wy[i] * I1wy[i];
            grad[i] = Ix2 + Iy2;
            rho_c[i] = I1w[i] - I1wx[i] * u1[i] - I1wy[i] * u2[i] - I0[i];
            ub1[i] = u1[i];
            ub2[i] = u2[i];
        }

    int it = 0;
    float err_D = INFINITY;
    while (err_D > tol_OF * tol_OF && it < max_iter_patch) {               /* :359 */
        it++;
        for (int l = ij; l < ej; l++)                                      /* :364-391 */
            for (int k = ii; k < ei; k++) {
                const int i = l * nx + k;
                const float rho = rho_c[i] + (I1wx[i] * u1[i] + I1wy[i] * u2[i]);
                const float l_t_w = weight ? l_t * weight[l - ij + ijw] * weight[k - ii + iiw] : l_t;   /* tvl2w :379 */
                float d1, d2;
                if (rho < -l_t_w * grad[i]) { d1 = l_t_w * I1wx[i]; d2 = l_t_w * I1wy[i]; }
                else if (rho > l_t_w * grad[i]) { d1 = -l_t_w * I1wx[i]; d2 = -l_t_w * I1wy[i]; }
                else if (grad[i] < 1E-8) { d1 = d2 = 0; }
                else { const float fi = -rho / grad[i]; d1 = fi * I1wx[i]; d2 = fi * I1wy[i]; }
                v1[i] = u1[i] + d1;
                v2[i] = u2[i] + d2;
            }
        fwd_grad_patch(ub1, u1x, u1y, ii, ij, ei, ej, nx);                 /* :396-397 */
        fwd_grad_patch(ub2, u2x, u2y, ii, ij, ei, ej, nx);
        for (int l = ij; l < ej; l++)                                      /* tvl2coupled_getD :82-115 */
            for (int k = ii; k < ei; k++) {
                const int i = l * nx + k;
                const float g11 = xi11[i] * xi11[i], g12 = xi12[i] * xi12[i], g21 = xi21[i] * xi21[i],
                            g22 = xi22[i] * xi22[i];
                float xi_N = sqrtf(g11 + g12 + g21 + g22);
                xi_N = (1 > xi_N) ? 1 : xi_N;
                xi11[i] = (xi11[i] + tau * u1x[i]) / xi_N;
                xi12[i] = (xi12[i] + tau * u1y[i]) / xi_N;
                xi21[i] = (xi21[i] + tau * u2x[i]) / xi_N;
                xi22[i] = (xi22[i] + tau * u2y[i]) / xi_N;
            }
        div_patch(xi11, xi12, div1, ii, ij, ei, ej, nx);                   /* :401-402 */
        div_patch(xi21, xi22, div2, ii, ij, ei, ej, nx);
        err_D = 0;                                                          /* tvl2coupled_getP :122-166 */
        for (int l = ij; l < ej; l++)
            for (int k = ii; k < ei; k++) {
                const int i = l * nx + k;
                const float u1k = u1[i], u2k = u2[i];
                u1[i] = u1k - tau * (-div1[i] + (u1k - v1[i]) / theta);
                u2[i] = u2k - tau * (-div2[i] + (u2k - v2[i]) / theta);
                const float uN = (u1[i] - u1k) * (u1[i] - u1k) + (u2[i] - u2k) * (u2[i] - u2k);
                if (err_D < uN) err_D = uN;
                ub1[i] = 2 * u1[i] - u1k;                                  /* :420-427 */
                ub2[i] = 2 * u2[i] - u2k;
            }
    }
    if (iters_out) *iters_out = it;

    /* eval_tvl2coupled :174-243 (v1, v2 of the last iteration; gradients and warp of the NEW flow) */
    fwd_grad_patch(u1, u1x, u1y, ii, ij, ei, ej, nx);
    fwd_grad_patch(u2, u2x, u2y, ii, ij, ei, ej, nx);
    float ener = 0.0f;
    int m = 0;
    for (int l = ij; l < ej; l++)
        for (int k = ii; k < ei; k++) {
            const int i = l * nx + k;
            const float iw = orc_bicubic_at(I1, (float)k + u1[i], (float)l + u2[i], nx, ny, 0);
            /* `fabs` resolves to the double function there, so the weighted product (tvl2w :227) runs in double
             * and is rounded once; without weights the single product rounds the same either way */
            const float dt = weight ? (float)((double)lambda * fabs((double)(iw - I0[i])) * weight[l - ij + ijw] * weight[k - ii + iiw])
                                    : lambda * fabsf(iw - I0[i]);
            const float dc = (1 / (2 * theta)) * ((u1[i] - v1[i]) * (u1[i] - v1[i]) + (u2[i] - v2[i]) * (u2[i] - v2[i]));
            const float g1 = u1x[i] * u1x[i], g12 = u1y[i] * u1y[i], g21 = u2x[i] * u2x[i], g2 = u2y[i] * u2y[i];
            const float g = sqrtf(g1 + g12 + g21 + g2);
            ener += dc + dt + g;
            m++;
        }
    ener = (float)(ener / (m * 1.0));
    return ener;
}

/* ------------------------------------------------------------------------------------------
 * Harmonic (Laplace) hole filling of a patch, src/elap_recsep.c:21-231, as interpolate_poisson
 * (src/local_faldoi.cpp:326-372) calls it: elap_recursive_separable(out, in, w, h, pd, 0.4, 3, 7).
 * Restated without recursion: a pyramid of up to `scale` levels is built top-down by 2x2 NaN-aware
 * block means, solved from the coarsest level (zero initialisation) upwards, each level started
 * from the pixel-replicated coarser solution.  The relaxation is sequential (Gauss-Seidel in mask
 * order): every update sees the neighbours already updated in the same sweep.
 * ---------------------------------------------------------------------------------------- */
#define ELAP_MAXPIX 2500           /* MAX_PATCH*MAX_PATCH, src/parameters.h:89 */

static float px_clamp(const float *x, int w, int h, int i, int j)      /* getpixel_1 :24-31 */
{
    if (i < 0) i = 0;
    if (j < 0) j = 0;
    if (i >= w) i = w - 1;
    if (j >= h) j = h - 1;
    return x[i + j * w];
}

static void harmonic_fill(float *y, const float *x, int w, int h, float tstep, int niter, const float *init)
{
    for (int i = 0; i < w * h; i++) y[i] = isfinite(x[i]) ? x[i] : init[i];        /* :113-114 */
    for (int it = 0; it < niter; it++) {                                            /* :117-126 */
        float maxupdate = 0;
        for (int j = 0; j < h; j++)                                                 /* mask order = row-major (:78-84) */
            for (int i = 0; i < w; i++) {
                if (!isnan(x[j * w + i])) continue;
                const int idx = j * w + i;
                const float lap = -4 * px_clamp(y, w, h, i, j) + px_clamp(y, w, h, i + 1, j) + px_clamp(y, w, h, i, j + 1)
                                  + px_clamp(y, w, h, i - 1, j) + px_clamp(y, w, h, i, j - 1);     /* :37-43 */
                const float nw = y[idx] + tstep * lap;
                const float update = (float)fabs(y[idx] - nw);
                if (update > maxupdate) maxupdate = update;
                y[idx] = nw;
            }
        if (maxupdate < 1e-10) break;
    }
}

void orc_elap_recursive(float *out, const float *in, int w, int h, float timestep, int niter, int scale)
{
    enum { MAXL = 16 };
    int ws[MAXL], hs[MAXL];
    float *ins[MAXL], *outs[MAXL];
    if (scale > MAXL) scale = MAXL;
    ws[0] = w; hs[0] = h;
    ins[0] = (float *)in;
    for (int s = 1; s < scale; s++) {                                               /* :195-201 */
        ws[s] = (int)ceil(ws[s - 1] / 2.0);
        hs[s] = (int)ceil(hs[s - 1] / 2.0);
        ins[s] = (float *)malloc(sizeof(float) * ws[s] * hs[s]);
        const float *src = ins[s - 1];
        const int iw = ws[s - 1], ih = hs[s - 1];
        for (int j = 0; j < hs[s]; j++)                                             /* zoom_out_by_factor_two :160-175 */
            for (int i = 0; i < ws[s]; i++) {
                const float a[4] = { px_clamp(src, iw, ih, 2 * i, 2 * j), px_clamp(src, iw, ih, 2 * i + 1, 2 * j),
                                     px_clamp(src, iw, ih, 2 * i, 2 * j + 1), px_clamp(src, iw, ih, 2 * i + 1, 2 * j + 1) };
                float m = 0;
                int cx = 0;
                for (int k = 0; k < 4; k++)
                    if (isfinite(a[k])) { m += a[k]; cx += 1; }
                ins[s][ws[s] * j + i] = cx ? m / cx : NAN;
            }
    }
    for (int s = scale - 1; s >= 0; s--) {
        const int n = ws[s] * hs[s];
        float *init = (float *)malloc(sizeof(float) * n);
        outs[s] = (s == 0) ? out : (float *)malloc(sizeof(float) * n);
        if (s == scale - 1) {
            for (int i = 0; i < n; i++) init[i] = 0;                                /* :204-206 */
        } else {                                                                    /* zoom_in_by_factor_two :181-189 */
            for (int j = 0; j < hs[s]; j++)
                for (int i = 0; i < ws[s]; i++)
                    init[ws[s] * j + i] = px_clamp(outs[s + 1], ws[s + 1], hs[s + 1], (int)round((i - 0.5) / 2),
                                                   (int)round((j - 0.5) / 2));
        }
        harmonic_fill(outs[s], ins[s], ws[s], hs[s], timestep, niter, init);        /* :208 */
        free(init);
    }
    for (int s = 1; s < scale; s++) { free(ins[s]); free(outs[s]); }
}

/* interpolate_poisson (src/local_faldoi.cpp:326-372): non-fixed pixels of the window become holes,
 * both flow channels are filled, ALL window pixels are written back. */
void orc_interpolate_poisson(float *u1, float *u2, const int *fixed_points, int nx,
                             int ii, int ij, int ei, int ej)
{
    const int w = ei - ii, h = ej - ij;
    float bin[2][ELAP_MAXPIX], bout[2][ELAP_MAXPIX];
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            const int xy = (j + ij) * nx + i + ii;
            const int fx = fixed_points[xy] == 1;
            bin[0][j * w + i] = fx ? u1[xy] : NAN;
            bin[1][j * w + i] = fx ? u2[xy] : NAN;
        }
    orc_elap_recursive(bout[0], bin[0], w, h, 0.4f, 3, 7);                          /* :357 */
    orc_elap_recursive(bout[1], bin[1], w, h, 0.4f, 3, 7);
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            const int xy = (j + ij) * nx + i + ii;
            u1[xy] = bout[0][j * w + i];
            u2[xy] = bout[1][j * w + i];
        }
}
