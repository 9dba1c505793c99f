// local_kernels.hpp -- the LOCAL TV-L1 patch operator on gfx950.
//
// Reference: of_estimation's default branch (src/energy_model.cpp:796-800) = guided_tvl2coupled
// (src/tvl2_model.cpp:249-435) + eval_tvl2coupled (:174-243): one outer warp and at most
// max_iter_patch (4) primal-dual iterations on a window [ii,ei) x [ij,ej) (<= 11x11) of full-size
// planes, with the patch stencils of src/utils.cpp:63-220, then the mean patch energy.
//
// One workgroup (128 threads, one per window pixel) per patch, any number of patches per launch.
// The reference operator is stateful and so is this one: u, the duals xi and BOTH divergence
// planes are persistent HBM planes.  divergence_patch leaves the window's first row and first /
// last column unwritten (it writes the IMAGE's row 0 / column 0 / column nx-1 instead, SURVEY
// App. B-6), so those entries keep what an earlier call left; the statement order of the
// reference is kept (barriers between its four statement groups).  A launch with several
// patches equals the sequential reference iff the windows (and those shadow writes) do not
// interact; the local growing of the reference is sequential, one patch per call.
#pragma once
#include "faldoi_device.hpp"

#pragma clang fp contract(off)

namespace faldoi {

constexpr int kPatchMax = 11;                    // 2*PAR_DEFAULT_WINSIZE + 1 (src/parameters.h:62)
constexpr int kPatchThreads = 128;

// planes of the local context: [L_COUNT][plane_elems]
enum LocalPlane { L_I0 = 0, L_I1, L_I1X, L_I1Y, L_U1, L_U2, L_X11, L_X12, L_X21, L_X22, L_DIV1, L_DIV2, L_COUNT };

struct PatchDesc { int ii, ij, ei, ej; };

__device__ __forceinline__ float bicubic_clamped(const float *img, float uu, float vv, int nx, int ny)
{
    const BicubicTaps t = bicubic_taps(uu, vv, nx, ny);      // border_out = false (src/tvl2_model.cpp:325-330)
    return bicubic_eval(img, t, nx);
}

// src/utils.cpp:63-113, one divergence plane; every thread of the workgroup calls it.
__device__ __forceinline__ void divergence_patch_dev(const float *v1, const float *v2, float *dv, int k, int l, bool act,
                                                     int ii, int ij, int ei, int ej, int nx)
{
    const int p = l * nx + k;
    if (act && k > ii && k < ei - 1 && l > ij && l < ej - 1)                       // :76-86 body
        dv[p] = (v1[p] - v1[p - 1]) + (v2[p] - v2[p - nx]);
    __syncthreads();
    if (act && k > ii && k < ei - 1) {                                             // :88-93
        if (l == ij) dv[k] = v1[k] - v1[k - 1] + v2[k];                            // IMAGE row 0 (sic)
        if (l == ej - 1) dv[p] = v1[p] - v1[p - 1] - v2[p - nx];                   // window's last row
    }
    __syncthreads();
    if (act && l > ij && l < ej - 1) {                                             // :96-103, IMAGE columns (sic)
        if (k == ii) { const int p1 = l * nx; dv[p1] = v1[p1] + v2[p1] - v2[p1 - nx]; }
        if (k == ei - 1) { const int p2 = (l + 1) * nx - 1; dv[p2] = -v1[p2 - 1] + v2[p2] - v2[p2 - nx]; }
    }
    __syncthreads();
    if (act) {                                                                     // :104-108 window corners
        if (k == ii && l == ij) dv[p] = v1[p] + v2[p];
        if (k == ei - 1 && l == ij) dv[p] = -v1[p - 1] + v2[p];
        if (k == ii && l == ej - 1) dv[p] = v1[p] - v2[p - nx];
        if (k == ei - 1 && l == ej - 1) dv[p] = -v1[p - 1] - v2[p - nx];
    }
    __syncthreads();
}

__global__ __launch_bounds__(kPatchThreads) void local_tvl1_patch_kernel(float *__restrict__ base, size_t pe, int nx, int ny,
                                                                       const PatchDesc *__restrict__ patches,
                                                                       int max_iter_patch, float *__restrict__ ener_out,
                                                                       int *__restrict__ iters_out)
{
    __shared__ float s_a[kPatchThreads], s_b[kPatchThreads];      // ubar (iterations) / u (energy) for neighbour reads
    __shared__ float s_t[kPatchThreads];                          // error terms, then energy terms
    const PatchDesc pd = patches[blockIdx.x];
    const int ii = pd.ii, ij = pd.ij, ei = pd.ei, ej = pd.ej;
    const int pw = ei - ii, ph = ej - ij, t = threadIdx.x;
    const bool act = t < pw * ph;
    const int k = ii + (act ? t % pw : 0), l = ij + (act ? t / pw : 0);
    const int p = l * nx + k;
    const float lambda = 40.0f, theta = 0.3f, tau = 0.125f, tol_OF = 0.01f;        // src/energy_model.cpp:704-709
    const float l_t = lambda * theta;
    const float *I0 = base + L_I0 * pe, *I1 = base + L_I1 * pe, *I1x = base + L_I1X * pe, *I1y = base + L_I1Y * pe;
    float *U1 = base + L_U1 * pe, *U2 = base + L_U2 * pe;
    float *X11 = base + L_X11 * pe, *X12 = base + L_X12 * pe, *X21 = base + L_X21 * pe, *X22 = base + L_X22 * pe;
    float *D1 = base + L_DIV1 * pe, *D2 = base + L_DIV2 * pe;

    float u1 = 0, u2 = 0, x11 = 0, x12 = 0, x21 = 0, x22 = 0, rc = 0, wx = 0, wy = 0, gr = 0, v1 = 0, v2 = 0, ub1 = 0, ub2 = 0;
    if (act) {
        u1 = U1[p];
        u2 = U2[p];
        X11[p] = 0.0f; X12[p] = 0.0f; X21[p] = 0.0f; X22[p] = 0.0f;               // src/tvl2_model.cpp:313-321
        const float uu = (float)k + u1, vv = (float)l + u2;                       // src/bicubic_interpolation.c:288-289
        const float w0 = bicubic_clamped(I1, uu, vv, nx, ny);                     // :325-330, warps = 1
        wx = bicubic_clamped(I1x, uu, vv, nx, ny);
        wy = bicubic_clamped(I1y, uu, vv, nx, ny);
        gr = grad_sq(wx, wy);                                                     // :336-346
        rc = rho_const(w0, wx, wy, u1, u2, I0[p]);
        ub1 = u1;                                                                 // :349-355
        ub2 = u2;
    }
    int it = 0;
    float err_D = __builtin_inff();
    while (err_D > tol_OF * tol_OF && it < max_iter_patch) {                      // :359 (uniform: err_D is shared)
        ++it;
        if (act) {                                                                // :364-391, plain IEEE quotient
            const float rho = rc + (wx * u1 + wy * u2);
            float d1, d2;
            if (rho < -l_t * gr) { d1 = l_t * wx; d2 = l_t * wy; }
            else if (rho > l_t * gr) { d1 = -l_t * wx; d2 = -l_t * wy; }
            else if (gr < grad_zero_bound()) { d1 = 0.0f; d2 = 0.0f; }
            else { const float fi = -rho / gr; d1 = fi * wx; d2 = fi * wy; }
            v1 = u1 + d1;
            v2 = u2 + d2;
        }
        s_a[t] = ub1;
        s_b[t] = ub2;
        __syncthreads();
        if (act) {
            // forward_gradient_patch (src/utils.cpp:175-220): zero on the WINDOW's last column / row
            const float u1x = (k < ei - 1) ? s_a[t + 1] - ub1 : 0.0f, u1y = (l < ej - 1) ? s_a[t + pw] - ub1 : 0.0f;
            const float u2x = (k < ei - 1) ? s_b[t + 1] - ub2 : 0.0f, u2y = (l < ej - 1) ? s_b[t + pw] - ub2 : 0.0f;
            const float g11 = x11 * x11, g12 = x12 * x12, g21 = x21 * x21, g22 = x22 * x22;   // tvl2coupled_getD :82-115
            float n = __builtin_sqrtf(g11 + g12 + g21 + g22);
            n = (1.0f > n) ? 1.0f : n;
            x11 = (x11 + tau * u1x) / n;
            x12 = (x12 + tau * u1y) / n;
            x21 = (x21 + tau * u2x) / n;
            x22 = (x22 + tau * u2y) / n;
            X11[p] = x11; X12[p] = x12; X21[p] = x21; X22[p] = x22;
        }
        __syncthreads();                                                          // duals visible to the workgroup
        divergence_patch_dev(X11, X12, D1, k, l, act, ii, ij, ei, ej, nx);        // :401-402
        divergence_patch_dev(X21, X22, D2, k, l, act, ii, ij, ei, ej, nx);
        float uN = 0.0f;
        if (act) {                                                                // tvl2coupled_getP :122-166
            const float u1k = u1, u2k = u2;
            u1 = u1k - tau * (-D1[p] + (u1k - v1) / theta);
            u2 = u2k - tau * (-D2[p] + (u2k - v2) / theta);
            uN = (u1 - u1k) * (u1 - u1k) + (u2 - u2k) * (u2 - u2k);
            ub1 = 2.0f * u1 - u1k;                                                // :420-427
            ub2 = 2.0f * u2 - u2k;
        }
        s_t[t] = uN;                                                              // max over the window (order-free)
        __syncthreads();
        for (int s = kPatchThreads / 2; s > 0; s >>= 1) {
            if (t < s) s_t[t] = s_t[t + s] > s_t[t] ? s_t[t + s] : s_t[t];
            __syncthreads();
        }
        err_D = s_t[0];
        __syncthreads();
    }

    // ---- eval_tvl2coupled (src/tvl2_model.cpp:174-243) ----
    s_a[t] = u1;
    s_b[t] = u2;
    __syncthreads();
    float term = 0.0f;
    if (act) {
        U1[p] = u1;
        U2[p] = u2;
        const float u1x = (k < ei - 1) ? s_a[t + 1] - u1 : 0.0f, u1y = (l < ej - 1) ? s_a[t + pw] - u1 : 0.0f;
        const float u2x = (k < ei - 1) ? s_b[t + 1] - u2 : 0.0f, u2y = (l < ej - 1) ? s_b[t + pw] - u2 : 0.0f;
        const float iw = bicubic_clamped(I1, (float)k + u1, (float)l + u2, nx, ny);
        const float dt = lambda * __builtin_fabsf(iw - I0[p]);
        const float dc = (1 / (2 * theta)) * ((u1 - v1) * (u1 - v1) + (u2 - v2) * (u2 - v2));
        const float g = __builtin_sqrtf(u1x * u1x + u1y * u1y + u2x * u2x + u2y * u2y);
        term = dc + dt + g;
    }
    s_t[t] = term;
    __syncthreads();
    if (t == 0) {                                        // the reference accumulates in row-major order, in fp32
        float ener = 0.0f;
        const int m = pw * ph;
        for (int q = 0; q < m; ++q) ener += s_t[q];
        ener_out[blockIdx.x] = (float)((double)ener / (m * 1.0));
        if (iters_out) iters_out[blockIdx.x] = it;
    }
}

// ---------------------------------------------------------------------------------------------
// interpolate_poisson (src/local_faldoi.cpp:326-372) = elap_recursive_separable(out, in, w, h, 2,
// 0.4, 3, 7) (src/elap_recsep.c:193-231) on a window: the non-fixed pixels become holes that are
// filled by a multiscale harmonic extension; every window pixel is written back.
// The relaxation of the reference is sequential (each update sees the neighbours already updated in
// the same sweep, src/elap_recsep.c:50-69), so one thread owns one (window, flow channel) and runs
// it in order; the parallelism is across windows.  The recursion is unrolled into a pyramid built
// top-down and solved bottom-up.
// ---------------------------------------------------------------------------------------------
constexpr int kElapLevels = 7;                   // `scale` of the call site
constexpr int kElapPix = kPatchMax * kPatchMax;  // 121
constexpr int kElapPyr = 121 + 36 + 9 + 4 + 1 + 1 + 1;   // 11 -> 6 -> 3 -> 2 -> 1 -> 1 -> 1

__device__ __forceinline__ float px_clamp(const float *x, int w, int h, int i, int j)     // getpixel_1 :24-31
{
    i = i < 0 ? 0 : (i >= w ? w - 1 : i);
    j = j < 0 ? 0 : (j >= h ? h - 1 : j);
    return x[i + j * w];
}

__global__ __launch_bounds__(64) void interpolate_poisson_kernel(float *__restrict__ base, size_t pe, int nx,
                                                               const int *__restrict__ fixed_points,
                                                               const PatchDesc *__restrict__ patches, int npatches)
{
    const int job = blockIdx.x * 64 + threadIdx.x;
    if (job >= 2 * npatches) return;
    const PatchDesc pd = patches[job >> 1];
    float *U = base + (size_t)((job & 1) ? L_U2 : L_U1) * pe;
    const int w0 = pd.ei - pd.ii, h0 = pd.ej - pd.ij;
    float in[kElapPyr], out[kElapPyr], init[kElapPix];
    int ws[kElapLevels], hs[kElapLevels], off[kElapLevels];
    ws[0] = w0; hs[0] = h0; off[0] = 0;
    for (int s = 1; s < kElapLevels; ++s) {                         // ceil(w/2.0), :195-196
        ws[s] = (ws[s - 1] + 1) / 2;
        hs[s] = (hs[s - 1] + 1) / 2;
        off[s] = off[s - 1] + ws[s - 1] * hs[s - 1];
    }
    for (int j = 0; j < h0; ++j)                                    // src/local_faldoi.cpp:341-355
        for (int i = 0; i < w0; ++i) {
            const int xy = (j + pd.ij) * nx + i + pd.ii;
            in[j * w0 + i] = (fixed_points[xy] == 1) ? U[xy] : __builtin_nanf("");
        }
    for (int s = 1; s < kElapLevels; ++s) {                         // zoom_out_by_factor_two :160-175
        const float *src = in + off[s - 1];
        float *dst = in + off[s];
        const int iw = ws[s - 1], ih = hs[s - 1];
        for (int j = 0; j < hs[s]; ++j)
            for (int i = 0; i < ws[s]; ++i) {
                const float a[4] = { px_clamp(src, iw, ih, 2 * i, 2 * j), px_clamp(src, iw, ih, 2 * i + 1, 2 * j),
                                     px_clamp(src, iw, ih, 2 * i, 2 * j + 1), px_clamp(src, iw, ih, 2 * i + 1, 2 * j + 1) };
                float m = 0;
                int cx = 0;
                for (int q = 0; q < 4; ++q)
                    if (__builtin_isfinite(a[q])) { m += a[q]; cx += 1; }
                dst[ws[s] * j + i] = cx ? m / cx : __builtin_nanf("");
            }
    }
    for (int s = kElapLevels - 1; s >= 0; --s) {
        const int w = ws[s], h = hs[s];
        const float *x = in + off[s];
        float *y = out + off[s];
        for (int j = 0; j < h; ++j)
            for (int i = 0; i < w; ++i) {
                float iv = 0.0f;                                    // coarsest level starts from zero, :204-206
                if (s < kElapLevels - 1)                            // zoom_in_by_factor_two :181-189
                    iv = px_clamp(out + off[s + 1], ws[s + 1], hs[s + 1], (int)__builtin_round((i - 0.5) / 2),
                                  (int)__builtin_round((j - 0.5) / 2));
                init[j * w + i] = iv;
            }
        for (int q = 0; q < w * h; ++q) y[q] = __builtin_isfinite(x[q]) ? x[q] : init[q];      // :113-114
        for (int it = 0; it < 3; ++it) {                            // niter = 3, tstep = 0.4
            float maxupdate = 0;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    if (!__builtin_isnan(x[j * w + i])) continue;
                    const int idx = j * w + i;
                    const float lap = -4 * px_clamp(y, w, h, i, j) + px_clamp(y, w, h, i + 1, j) + px_clamp(y, w, h, i, j + 1)
                                      + px_clamp(y, w, h, i - 1, j) + px_clamp(y, w, h, i, j - 1);
                    const float nw = y[idx] + 0.4f * lap;
                    const float update = __builtin_fabsf(y[idx] - nw);
                    if (update > maxupdate) maxupdate = update;
                    y[idx] = nw;
                }
            if ((double)maxupdate < 1e-10) break;
        }
    }
    for (int j = 0; j < h0; ++j)                                    // src/local_faldoi.cpp:359-368
        for (int i = 0; i < w0; ++i) U[(j + pd.ij) * nx + i + pd.ii] = out[j * w0 + i];
}

}  // namespace faldoi
