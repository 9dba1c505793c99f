/*
 * faldoi_oracle.c -- CPU restatement of the FALDOI TV-L1 hot path (see faldoi_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path and the timed CPU baseline of
 * bench.py.  Never linked into the product.
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off -fopenmp -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off is load-bearing: the reference is compiled without FMA contraction
 * (src/CMakeLists.txt:21-22), and a fused multiply-add changes the low bits.
 *
 * All citations are file:line in the upstream repo (fperezgamonal/faldoi-ipol).
 */
#include "faldoi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Stencils.  The reference walks body / edge rows / edge columns / corners in separate loops
 * (src/utils.cpp:239-423); here one loop classifies each pixel.  The expression written for
 * each class is the reference's, association included (it matters for the last bit).
 * ---------------------------------------------------------------------------------------- */

/* src/utils.cpp:367-423 (== src/mask.c:184): half the two-point difference; one-sided (still
 * halved) on the image frame.  The subtraction is float, the 0.5* is double, stored as float. */
void orc_centered_gradient(const float *in, float *dx, float *dy, int nx, int ny)
{
#pragma omp parallel for
    for (int y = 0; y < ny; y++) {
        const int yu = y > 0 ? y - 1 : y;
        const int yd = y < ny - 1 ? y + 1 : y;
        for (int x = 0; x < nx; x++) {
            const int xl = x > 0 ? x - 1 : x;
            const int xr = x < nx - 1 ? x + 1 : x;
            const float ddx = in[y * nx + xr] - in[y * nx + xl];
            const float ddy = in[yd * nx + x] - in[yu * nx + x];
            dx[y * nx + x] = (float)(0.5 * ddx);
            dy[y * nx + x] = (float)(0.5 * ddy);
        }
    }
}

/* src/utils.cpp:285-323: f[p+1]-f[p], f[p+nx]-f[p]; zero on the last column / last row. */
void orc_forward_gradient(const float *f, float *fx, float *fy, int nx, int ny)
{
#pragma omp parallel for
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            const int p = y * nx + x;
            fx[p] = (x < nx - 1) ? f[p + 1] - f[p] : 0.0f;
            fy[p] = (y < ny - 1) ? f[p + nx] - f[p] : 0.0f;
        }
}

/* One divergence sample with the reference's border rules (src/utils.cpp:239-283).
 * v1l = v1[p-1], v2u = v2[p-nx] (only read where the rule uses them). */
static inline float div_at(const float *v1, const float *v2, int x, int y, int nx, int ny)
{
    const int p = y * nx + x;
    const int first_c = (x == 0), last_c = (x == nx - 1);
    const int first_r = (y == 0), last_r = (y == ny - 1);

    if (!first_c && !last_c) {
        const float v1x = v1[p] - v1[p - 1];
        if (first_r) return v1x + v2[p];                 /* :263 */
        if (last_r) return v1x - v2[p - nx];             /* :264 */
        return v1x + (v2[p] - v2[p - nx]);               /* :250-256 */
    }
    if (first_c) {
        if (first_r) return v1[p] + v2[p];               /* :278 */
        if (last_r) return v1[p] - v2[p - nx];           /* :280 */
        return v1[p] + v2[p] - v2[p - nx];               /* :272 */
    }
    /* last column */
    if (first_r) return -v1[p - 1] + v2[p];              /* :279 */
    if (last_r) return -v1[p - 1] - v2[p - nx];          /* :281 */
    return -v1[p - 1] + v2[p] - v2[p - nx];              /* :273 */
}

void orc_divergence(const float *v1, const float *v2, float *div, int nx, int ny)
{
#pragma omp parallel for
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++)
            div[y * nx + x] = div_at(v1, v2, x, y, nx, ny);
}

/* ------------------------------------------------------------------------------------------
 * Bicubic interpolation (src/bicubic_interpolation.c).
 * ---------------------------------------------------------------------------------------- */

/* :26-40  Neumann clamp that also records that clamping happened. */
static inline int clamp_flag(int i, int n, int *clamped)
{
    if (i < 0) { *clamped = 1; return 0; }
    if (i >= n) { *clamped = 1; return n - 1; }
    return i;
}

/* :103-111  Keys cubic through 4 samples at parameter t.  Mixed precision exactly as C
 * promotes the reference expression: (v2-v0) and (v1-v2) are float subtractions, everything
 * touching a double literal runs in double, the result is rounded to float once. */
static inline float keys_cubic(float v0, float v1, float v2, float v3, float t)
{
    const float d20 = v2 - v0;
    const float d12 = v1 - v2;
    const double td = (double)t;
    const double c3 = 3.0 * (double)d12 + (double)v3 - (double)v0;
    const double c2 = 2.0 * (double)v0 - 5.0 * (double)v1 + 4.0 * (double)v2 - (double)v3 + td * c3;
    const double c1 = (double)d20 + td * c2;
    return (float)((double)v1 + 0.5 * td * c1);
}

/* :138-237 */
float orc_bicubic_at(const float *in, float uu, float vv, int nx, int ny, int border_out)
{
    const int sx = (uu < 0) ? -1 : 1;
    const int sy = (vv < 0) ? -1 : 1;
    const int iu = (int)uu, iv = (int)vv;       /* truncation toward zero */
    int clamped = 0;

    /* tap columns / rows in the reference's order (centre, minus, plus, plus-plus).
     * NB the "minus" row uses sx, not sy -- reference quirk at :159, reproduced. */
    const int x = clamp_flag(iu, nx, &clamped);
    const int y = clamp_flag(iv, ny, &clamped);
    const int mx = clamp_flag(iu - sx, nx, &clamped);
    const int my = clamp_flag(iv - sx, ny, &clamped);
    const int dx = clamp_flag(iu + sx, nx, &clamped);
    const int dy = clamp_flag(iv + sy, ny, &clamped);
    const int ddx = clamp_flag(iu + 2 * sx, nx, &clamped);
    const int ddy = clamp_flag(iv + 2 * sy, ny, &clamped);

    if (clamped && border_out) return 0.0f;

    const int X[4] = { mx, x, dx, ddx };
    const int Y[4] = { my, y, dy, ddy };
    const float ty = vv - (float)y;             /* float - int -> float */
    const float tx = uu - (float)x;
    float col[4];
    for (int a = 0; a < 4; a++)                 /* :119-130: first along y for each tap column */
        col[a] = keys_cubic(in[X[a] + nx * Y[0]], in[X[a] + nx * Y[1]],
                            in[X[a] + nx * Y[2]], in[X[a] + nx * Y[3]], ty);
    return keys_cubic(col[0], col[1], col[2], col[3], tx);
}

/* :245-266 */
void orc_bicubic_warp(const float *in, const float *u, const float *v, float *out,
                      int nx, int ny, int border_out)
{
#pragma omp parallel for
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            const int p = y * nx + x;
            const float uu = (float)x + u[p];
            const float vv = (float)y + v[p];
            out[p] = orc_bicubic_at(in, uu, vv, nx, ny, border_out);
        }
}

/* ------------------------------------------------------------------------------------------
 * Preprocessing.
 * ---------------------------------------------------------------------------------------- */

/* src/global_faldoi.cpp:1820-1828: double weights, one rounding to float. */
void orc_rgb2gray(const float *in, int w, int h, float *out)
{
    const int n = w * h;
    for (int i = 0; i < n; i++)
        out[i] = (float)(.299 * in[i] + .587 * in[n + i] + .114 * in[2 * n + i]);
}

/* src/utils.cpp:679-693 */
void orc_getminmax(float *mn, float *mx, const float *x, int n)
{
    float lo = x[0], hi = x[0];
    for (int i = 1; i < n; i++) {
        if (x[i] < lo) lo = x[i];
        if (x[i] > hi) hi = x[i];
    }
    *mn = lo;
    *mx = hi;
}

/* src/utils.cpp:743-781.  The "min" it uses is the LARGER of (min of third image) and
 * (smaller min of the first two) -- reference quirk at :763, reproduced.  Safe in place. */
void orc_image_normalization_3(const float *I1, const float *I2, const float *I0,
                               float *I1n, float *I2n, float *I0n, int size)
{
    float max0, max1, max2, min0, min1, min2;
    orc_getminmax(&min0, &max0, I0, size);
    orc_getminmax(&min1, &max1, I1, size);
    orc_getminmax(&min2, &max2, I2, size);
    const float max01 = (max0 > max1) ? max0 : max1;
    const float max = (max2 > max01) ? max2 : max01;
    const float min01 = (min0 < min1) ? min0 : min1;
    const float min = (min2 > min01) ? min2 : min01;
    const float den = max - min;
    for (int i = 0; i < size; i++) {
        const float a = I0[i], b = I1[i], c = I2[i];
        I0n[i] = (den > 0) ? (a - min) / den : a;
        I1n[i] = (den > 0) ? (b - min) / den : b;
        I2n[i] = (den > 0) ? (c - min) / den : c;
    }
}

/* src/utils.cpp:703-735 (two-image version used by the local step's prepare_stuff). */
void orc_image_normalization_2(const float *I0, const float *I1, float *I0n, float *I1n, int size)
{
    float max0, max1, min0, min1;
    orc_getminmax(&min0, &max0, I0, size);
    orc_getminmax(&min1, &max1, I1, size);
    const float max = (max0 > max1) ? max0 : max1;
    const float min = (min0 < min1) ? min0 : min1;
    const float den = max - min;
    for (int i = 0; i < size; i++) {
        const float a = I0[i], b = I1[i];
        I0n[i] = (den > 0) ? (a - min) / den : a;
        I1n[i] = (den > 0) ? (b - min) / den : b;
    }
}

/* Gaussian taps as src/utils.cpp:529-552 computes them in its C++ translation unit:
 * exp() of a float argument resolves to the float overload (expf); the prefactor and the
 * product are double; the normalisation accumulates in float. */
int orc_gaussian_taps(float sigma, float *B, int maxn)
{
    const float den = 2 * sigma * sigma;
    const int size = (int)(5 * sigma) + 1;
    if (size > maxn) return -1;
    for (int i = 0; i < size; i++) {
        const float arg = -i * i / den;
        B[i] = (float)(1 / (sigma * sqrt(2.0 * 3.1415926)) * (double)expf(arg));
    }
    float norm = 0;
    for (int i = 0; i < size; i++) norm += B[i];
    norm *= 2;
    norm -= B[0];
    for (int i = 0; i < size; i++) B[i] /= norm;
    return size;
}

/* The C copy of the Gaussian (src/mask.c:248-340, the one zoom.c links against): identical text,
 * but in C `exp` is the double function, so the exponent is evaluated in double. */
int orc_gaussian_taps_c(float sigma, float *B, int maxn)
{
    const float den = 2 * sigma * sigma;
    const int size = (int)(5 * sigma) + 1;
    if (size > maxn) return -1;
    for (int i = 0; i < size; i++)
        B[i] = (float)(1 / (sigma * sqrt(2.0 * 3.1415926)) * exp(-i * i / den));
    float norm = 0;
    for (int i = 0; i < size; i++) norm += B[i];
    norm *= 2;
    norm -= B[0];
    for (int i = 0; i < size; i++) B[i] /= norm;
    return size;
}

/* Index of the sample the reference's padded line buffer holds at virtual position i
 * (src/utils.cpp:566-571,600-605, "reflecting"): the left/top pad mirrors WITHOUT repeating
 * the edge sample, the right/bottom pad mirrors WITH it. */
static inline int reflect_index(int i, int n)
{
    if (i < 0) return -i;
    if (i >= n) return 2 * n - i - 1;
    return i;
}

static void gaussian_line(const float *src, int n, int stride, const float *B, int size, float *dst)
{
    for (int i = 0; i < n; i++) {
        float sum = B[0] * src[i * stride];
        for (int j = 1; j < size; j++) {
            const float a = src[reflect_index(i - j, n) * stride];
            const float b = src[reflect_index(i + j, n) * stride];
            sum += B[j] * (a + b);
        }
        dst[i] = sum;
    }
}

static void gaussian_with_taps(float *I, int xdim, int ydim, const float *B, int size);

/* src/utils.cpp:521-630: separable, rows first, then columns on the row-filtered image. */
void orc_gaussian(float *I, int xdim, int ydim, float sigma)
{
    float B[64];
    const int size = orc_gaussian_taps(sigma, B, 64);
    if (size < 0 || size > xdim || size >= ydim + 1) abort();
    gaussian_with_taps(I, xdim, ydim, B, size);
}

/* src/mask.c:248 (C linkage copy) */
void orc_gaussian_c(float *I, int xdim, int ydim, float sigma)
{
    float B[64];
    const int size = orc_gaussian_taps_c(sigma, B, 64);
    if (size < 0 || size > xdim || size >= ydim + 1) abort();
    gaussian_with_taps(I, xdim, ydim, B, size);
}

static void gaussian_with_taps(float *I, int xdim, int ydim, const float *B, int size)
{

#pragma omp parallel
    {
        float *line = (float *)malloc(sizeof(float) * (size_t)(xdim > ydim ? xdim : ydim));
#pragma omp for
        for (int k = 0; k < ydim; k++) {
            gaussian_line(I + (size_t)k * xdim, xdim, 1, B, size, line);
            memcpy(I + (size_t)k * xdim, line, sizeof(float) * (size_t)xdim);
        }
#pragma omp for
        for (int k = 0; k < xdim; k++) {
            gaussian_line(I + k, ydim, xdim, B, size, line);
            for (int i = 0; i < ydim; i++) I[(size_t)i * xdim + k] = line[i];
        }
        free(line);
    }
}

/* src/global_faldoi.cpp:2049-2068 */
void orc_preprocess_global(const float *i0, const float *i1, const float *im1, int w, int h, int pd,
                           float *i0n, float *i1n, float *im1n)
{
    const int n = w * h;
    if (pd != 1) {
        orc_rgb2gray(i0, w, h, i0n);
        orc_rgb2gray(i1, w, h, i1n);
        orc_rgb2gray(im1, w, h, im1n);
    } else {
        memcpy(i0n, i0, sizeof(float) * (size_t)n);
        memcpy(i1n, i1, sizeof(float) * (size_t)n);
        memcpy(im1n, im1, sizeof(float) * (size_t)n);
    }
    /* call-site binding at :2065: (I1:=i0n, I2:=i1n, I0:=i_1n) */
    orc_image_normalization_3(i0n, i1n, im1n, i0n, i1n, im1n, n);
    orc_gaussian(i0n, w, h, 0.90f);
    orc_gaussian(i1n, w, h, 0.90f);
    orc_gaussian(im1n, w, h, 0.90f);
}

/* ------------------------------------------------------------------------------------------
 * TV-L1 primal-dual solver (tvl2OF).
 * ---------------------------------------------------------------------------------------- */

/* src/global_faldoi.cpp:635-657 */
void orc_tvl1_warp_constants(const float *I0, const float *I1, const float *I1x, const float *I1y,
                             const float *u1, const float *u2,
                             float *I1wx, float *I1wy, float *grad, float *rho_c, int nx, int ny)
{
#pragma omp parallel for
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            const int p = y * nx + x;
            const float uu = (float)x + u1[p];
            const float vv = (float)y + u2[p];
            const float w0 = orc_bicubic_at(I1, uu, vv, nx, ny, 1);
            const float wx = orc_bicubic_at(I1x, uu, vv, nx, ny, 1);
            const float wy = orc_bicubic_at(I1y, uu, vv, nx, ny, 1);
            const float Ix2 = wx * wx;
            const float Iy2 = wy * wy;
            I1wx[p] = wx;
            I1wy[p] = wy;
            grad[p] = Ix2 + Iy2;
            rho_c[p] = w0 - wx * u1[p] - wy * u2[p] - I0[p];
        }
}

/* One primal-dual iteration, as two fused sweeps instead of the reference's seven passes
 * (src/global_faldoi.cpp:690-783).  Sweep A: forward gradient of ubar (:725-726) + dual step
 * (:349-381).  Sweep B: divergence of the NEW duals (:745-746) + thresholding step (:690-718)
 * + primal step and max |du|^2 (:307-342) + over-relaxation (:780-783).  Every value a pixel
 * reads from a neighbour belongs to the same time level as in the reference's pass order. */
float orc_tvl1_iteration(const float *rho_c, const float *I1wx, const float *I1wy, const float *grad,
                         float *u1, float *u2, float *ub1, float *ub2,
                         float *xi11, float *xi12, float *xi21, float *xi22,
                         float lambda, float theta, float tau, int nx, int ny)
{
    const float l_t = lambda * theta;
    float err = 0.0f;

#pragma omp parallel for
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            const int p = y * nx + x;
            const float u1x = (x < nx - 1) ? ub1[p + 1] - ub1[p] : 0.0f;
            const float u1y = (y < ny - 1) ? ub1[p + nx] - ub1[p] : 0.0f;
            const float u2x = (x < nx - 1) ? ub2[p + 1] - ub2[p] : 0.0f;
            const float u2y = (y < ny - 1) ? ub2[p + nx] - ub2[p] : 0.0f;

            const float g11 = xi11[p] * xi11[p];
            const float g12 = xi12[p] * xi12[p];
            const float g21 = xi21[p] * xi21[p];
            const float g22 = xi22[p] * xi22[p];
            float xi_N = sqrtf(g11 + g12 + g21 + g22);
            xi_N = (1 > xi_N) ? 1 : xi_N;

            xi11[p] = (xi11[p] + tau * u1x) / xi_N;
            xi12[p] = (xi12[p] + tau * u1y) / xi_N;
            xi21[p] = (xi21[p] + tau * u2x) / xi_N;
            xi22[p] = (xi22[p] + tau * u2y) / xi_N;
        }

#pragma omp parallel for reduction(max : err)
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            const int p = y * nx + x;
            const float div1 = div_at(xi11, xi12, x, y, nx, ny);
            const float div2 = div_at(xi21, xi22, x, y, nx, ny);

            const float u1k = u1[p], u2k = u2[p];
            const float rho = rho_c[p] + (I1wx[p] * u1k + I1wy[p] * u2k);
            float d1, d2;
            if (rho < -l_t * grad[p]) {
                d1 = l_t * I1wx[p];
                d2 = l_t * I1wy[p];
            } else if (rho > l_t * grad[p]) {
                d1 = -l_t * I1wx[p];
                d2 = -l_t * I1wy[p];
            } else if (grad[p] < 1E-8) {       /* float promoted to double for the compare */
                d1 = d2 = 0;
            } else {
                const float fi = -rho / grad[p];
                d1 = fi * I1wx[p];
                d2 = fi * I1wy[p];
            }
            const float v1 = u1k + d1;
            const float v2 = u2k + d2;

            const float n1 = u1k - tau * (-div1 + (u1k - v1) / theta);
            const float n2 = u2k - tau * (-div2 + (u2k - v2) / theta);
            const float uN = (n1 - u1k) * (n1 - u1k) + (n2 - u2k) * (n2 - u2k);
            u1[p] = n1;
            u2[p] = n2;
            ub1[p] = 2 * n1 - u1k;
            ub2[p] = 2 * n2 - u2k;
            /* getminmax (src/utils.cpp:679) seeds with u_N[0] and replaces on '>'; u_N >= 0, so
             * seeding with 0 gives the same maximum for every NaN-free input. */
            if (uN > err) err = uN;
        }
    return err;
}

/* src/global_faldoi.cpp:556-882 */
long orc_tvl1_global(const float *I0, const float *I1, float *u1, float *u2,
                     float *xi11, float *xi12, float *xi21, float *xi22,
                     float lambda, float theta, float tau, float tol,
                     int nx, int ny, int warps, int max_inner, int tol_stop,
                     int *iters_out, float *err_out)
{
    const size_t n = (size_t)nx * (size_t)ny;
    float *buf = (float *)malloc(sizeof(float) * n * 8);
    if (!buf) return -1;
    float *I1x = buf, *I1y = buf + n, *I1wx = buf + 2 * n, *I1wy = buf + 3 * n;
    float *grad = buf + 4 * n, *rho_c = buf + 5 * n, *ub1 = buf + 6 * n, *ub2 = buf + 7 * n;
    long total = 0;

    orc_centered_gradient(I1, I1x, I1y, nx, ny);                       /* :611 */

    for (int w = 0; w < warps; w++) {                                  /* :631 */
        orc_tvl1_warp_constants(I0, I1, I1x, I1y, u1, u2, I1wx, I1wy, grad, rho_c, nx, ny);
        memcpy(ub1, u1, sizeof(float) * n);                            /* :659-660 */
        memcpy(ub2, u2, sizeof(float) * n);

        int it = 0;
        float err = INFINITY;                                          /* :683 */
        while ((!tol_stop || err > tol * tol) && it < max_inner) {     /* :684 */
            it++;
            err = orc_tvl1_iteration(rho_c, I1wx, I1wy, grad, u1, u2, ub1, ub2,
                                     xi11, xi12, xi21, xi22, lambda, theta, tau, nx, ny);
        }
        if (iters_out) iters_out[w] = it;
        if (err_out) err_out[w] = err;
        total += it;
    }
    free(buf);
    return total;
}

/* ------------------------------------------------------------------------------------------
 * Pyramid resampling, src/zoom.c (not compiled into the reference's executables; restated because
 * the north star names it).
 * ---------------------------------------------------------------------------------------- */

/* src/zoom.c:19-31 */
void orc_zoom_size(int nx, int ny, int *nxx, int *nyy, float factor)
{
    *nxx = (int)((float)nx * factor + 0.5);
    *nyy = (int)((float)ny * factor + 0.5);
}

/* src/zoom.c:38-76: Gaussian (mask.c copy) with sigma = 0.6*sqrt(1/f^2 - 1), then bicubic samples at
 * (j/f, i/f) with clamped taps (border_out = false). */
void orc_zoom_out(const float *I, float *Iout, int nx, int ny, float factor)
{
    float *Is = (float *)malloc(sizeof(float) * (size_t)nx * ny);
    memcpy(Is, I, sizeof(float) * (size_t)nx * ny);
    int nxx, nyy;
    orc_zoom_size(nx, ny, &nxx, &nyy, factor);
    const float sigma = (float)(0.6 * sqrt(1.0 / (factor * factor) - 1.0));
    orc_gaussian_c(Is, nx, ny, sigma);
#pragma omp parallel for
    for (int i1 = 0; i1 < nyy; i1++)
        for (int j1 = 0; j1 < nxx; j1++) {
            const float i2 = (float)i1 / factor;
            const float j2 = (float)j1 / factor;
            Iout[i1 * nxx + j1] = orc_bicubic_at(Is, j2, i2, nx, ny, 0);
        }
    free(Is);
}

/* src/zoom.c:82-106 */
void orc_zoom_in(const float *I, float *Iout, int nx, int ny, int nxx, int nyy)
{
    const float factorx = ((float)nxx / nx);
    const float factory = ((float)nyy / ny);
#pragma omp parallel for
    for (int i1 = 0; i1 < nyy; i1++)
        for (int j1 = 0; j1 < nxx; j1++) {
            const float i2 = (float)i1 / factory;
            const float j2 = (float)j1 / factorx;
            Iout[i1 * nxx + j1] = orc_bicubic_at(I, j2, i2, nx, ny, 0);
        }
}
