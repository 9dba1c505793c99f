// nltv_kernels.hpp -- NLTV-CSAD global minimisation (BASELINE configs[4]) on gfx950.
//
// Reference: nltvcsad_PD, src/global_faldoi.cpp:1642-1808, with ofnltv_getD :1127, non_local_divergence
// :1056, ofnltv_getP :1090, initialize_dual_variables :996, initialize_pos_nei :1332.
//
// Layout: everything is SoA planes of nx*ny floats.  The reference's per-pixel AoS tables
// (DualVariables_global: sc/wp/ap/rp[24] + wt; PosNei: api/apj/b[48] + a std::vector) become
//   p[24], q[24]   dual values            w[24], wt   weights (host-computed, libm)      b[48]  CSAD terms
// and the index tables disappear: slot s of the (2r+1)^2-1 window is the fixed offset
// (k,l) = rows -r..r, columns -r..r, centre skipped; a slot is valid iff the neighbour lies in the
// image; the mirrored slot of s is n_slots-1-s (the reference's rp, :1033).
// Per-pixel state is ~0.7 KB, far beyond registers, so unlike TV-L1 this solver streams: two
// launches per inner iteration (dual step; data + divergence + primal step), neighbours' duals are
// read as shifted plane rows (coalesced) -- the "gather stress" the config names.
#pragma once
#include "faldoi_device.hpp"

#pragma clang fp contract(off)

namespace faldoi {

constexpr int kNlR = 2, kNlSlots = 24;     // MAX_BETA / MAX_DUAL_VAR, src/global_faldoi.cpp:887-888
constexpr int kCsR = 3, kCsSlots = 48;     // DT_R / DT_NEI, src/parameters.h:86-87

// Plane indices of one NLTV-CSAD problem
enum NlPlane {
    N_I0 = 0, N_I1, N_I1X, N_I1Y, N_I1W, N_I1WX, N_I1WY, N_GRAD, N_U1, N_U2, N_UB1, N_UB2, N_WT,
    N_P,                                   // 24 planes
    N_Q = N_P + kNlSlots,                  // 24 planes
    N_W = N_Q + kNlSlots,                  // 24 planes
    N_B = N_W + kNlSlots,                  // 48 planes
    N_COUNT = N_B + kCsSlots
};

// Per-warp: three bicubic warps (border_out = true), grad = I1wx^2 + I1wy^2, ubar = u (:1694-1725)
// data_l1 (NLTV-L1, src/global_faldoi.cpp:1226-1234): the N_I1W plane receives rho_c instead of I1w.
__global__ __launch_bounds__(256) void nl_warp_kernel(float *__restrict__ base, size_t pe, int nx, int ny, int data_l1)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= nx * ny) return;
    const int y = p / nx, x = p - y * nx;
    const float u1 = base[N_U1 * pe + p], u2 = base[N_U2 * pe + p];
    const BicubicTaps tp = bicubic_taps((float)x + u1, (float)y + u2, nx, ny);
    float w0 = 0.0f, wx = 0.0f, wy = 0.0f;
    if (!tp.clamped) {
        w0 = bicubic_eval(base + N_I1 * pe, tp, nx);
        wx = bicubic_eval(base + N_I1X * pe, tp, nx);
        wy = bicubic_eval(base + N_I1Y * pe, tp, nx);
    }
    base[N_I1W * pe + p] = data_l1 ? rho_const(w0, wx, wy, u1, u2, base[N_I0 * pe + p]) : w0;
    base[N_I1WX * pe + p] = wx;
    base[N_I1WY * pe + p] = wy;
    base[N_GRAD * pe + p] = grad_sq(wx, wy);
    base[N_UB1 * pe + p] = u1;
    base[N_UB2 * pe + p] = u2;
}

// Per-warp: CSAD terms b_s for the 48 neighbours, only where grad > GRAD_IS_ZERO; elsewhere the
// previous value stays (:1704-1719).  grad > 1E-8 (double) <=> !(grad < grad_zero_bound()).
__global__ __launch_bounds__(256) void nl_csad_terms_kernel(float *__restrict__ base, size_t pe, int nx, int ny)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx * ny) return;
    const int y = i / nx, x = i - y * nx;
    const float grad = base[N_GRAD * pe + i];
    if (grad < grad_zero_bound()) return;
    const float *I0 = base + N_I0 * pe, *I1w = base + N_I1W * pe;
    const float i0 = I0[i], i1w = I1w[i];
    const float t1 = base[N_I1WX * pe + i] * base[N_U1 * pe + i];
    const float t2 = base[N_I1WY * pe + i] * base[N_U2 * pe + i];
    const float sg = __builtin_sqrtf(grad);
    int s = 0;
#pragma unroll
    for (int k = -kCsR; k <= kCsR; ++k)
#pragma unroll
        for (int l = -kCsR; l <= kCsR; ++l) {
            if (k == 0 && l == 0) continue;
            if (x + l >= 0 && x + l < nx && y + k >= 0 && y + k < ny) {
                const int pos = i + k * nx + l;
                base[(size_t)(N_B + s) * pe + i] = (i0 - I0[pos] - i1w + I1w[pos] + t1 + t2) / sg;   // :1716-1717
            }
            ++s;
        }
}

// ofnltv_getD (:1127-1174) on ubar: for every valid slot
//   g = w * (ubar_i - ubar_j) / wt;  p <- (p + tau*g) / (1 + tau*sqrt(g*g))     (same for q with ubar2)
__global__ __launch_bounds__(256) void nl_dual_kernel(float *__restrict__ base, size_t pe, int nx, int ny, float tau)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx * ny) return;
    const int y = i / nx, x = i - y * nx;
    const float *ub1 = base + N_UB1 * pe, *ub2 = base + N_UB2 * pe;
    const float a1 = ub1[i], a2 = ub2[i], wt = base[N_WT * pe + i];
    int s = 0;
#pragma unroll
    for (int k = -kNlR; k <= kNlR; ++k)
#pragma unroll
        for (int l = -kNlR; l <= kNlR; ++l) {
            if (k == 0 && l == 0) continue;
            if (x + l >= 0 && x + l < nx && y + k >= 0 && y + k < ny) {
                const int ap = i + k * nx + l;
                const float w = base[(size_t)(N_W + s) * pe + i];
                float *pp = base + (size_t)(N_P + s) * pe + i, *qq = base + (size_t)(N_Q + s) * pe + i;
                const float g1 = w * (a1 - ub1[ap]) / wt;
                const float d1 = 1 + tau * __builtin_sqrtf(g1 * g1);
                *pp = (*pp + tau * g1) / d1;
                const float g2 = w * (a2 - ub2[ap]) / wt;
                const float d2 = 1 + tau * __builtin_sqrtf(g2 * g2);
                *qq = (*qq + tau * g2) / d2;
            }
            ++s;
        }
}

// Compare-exchange network: bitonic sort of 64 registers, ascending (static indices only).
__device__ __forceinline__ void cswap(float &a, float &b)
{
    const float lo = __builtin_fminf(a, b);   // v_min_f32 / v_max_f32; -0 orders before +0, which
    const float hi = __builtin_fmaxf(a, b);   // std::sort leaves unspecified
    a = lo;
    b = hi;
}
__device__ __forceinline__ void bitonic_sort64(float (&v)[64])
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    if ((i & k) == 0) cswap(v[i], v[l]);
                    else cswap(v[l], v[i]);
                }
            }
}

// Data step (:1733-1758) + non-local divergence (:1056-1079) + ofnltv_getP (:1090-1120) + relaxation.
// The data step needs the element of index n+1 (0-based; `it/2+1` with it = 2n+1, sic) of the
// ascending sort of  A = { -(b_s - r) : valid s }  united with  B = { (n-2j)*l_t*sg : j = 0..n }.
// A is sorted in registers (padding = +inf); B is an arithmetic progression, so the order statistic
// is  min over i of max(A_sorted[i-1], B_ascending[n+1-i])  -- the classic two-sorted-arrays split.
// DATA_L1 = true: the data step is TV-L1's thresholding (nltvl1_PD, :1246-1272) on rho_c.
template <bool DATA_L1>
__global__ __launch_bounds__(256) void nl_primal_kernel(float *__restrict__ base, size_t pe, int nx, int ny,
                                                        float l_t, float theta, float tau)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx * ny) return;
    const int y = i / nx, x = i - y * nx;
    const float u1k = base[N_U1 * pe + i], u2k = base[N_U2 * pe + i];
    const float grad = base[N_GRAD * pe + i];
    float v1 = u1k, v2 = u2k;                                            // :1735-1736
    if (DATA_L1) {
        const float ixw = base[N_I1WX * pe + i], iyw = base[N_I1WY * pe + i];
        threshold_step(base[N_I1W * pe + i], ixw, iyw, grad, recip_f64(grad), u1k, u2k, l_t, v1, v2);
    } else if (!(grad < grad_zero_bound())) {
        const float ixw = base[N_I1WX * pe + i], iyw = base[N_I1WY * pe + i];
        const float sg = __builtin_sqrtf(grad);
        const float r = (ixw * u1k + iyw * u2k) / sg;
        float A[64];
        int n = 0, s = 0;
#pragma unroll
        for (int k = -kCsR; k <= kCsR; ++k)
#pragma unroll
            for (int l = -kCsR; l <= kCsR; ++l) {
                if (k == 0 && l == 0) continue;
                const bool valid = x + l >= 0 && x + l < nx && y + k >= 0 && y + k < ny;
                const float bs = base[(size_t)(N_B + s) * pe + i];
                A[s] = valid ? -(bs - r) : __builtin_inff();            // :1741-1742
                n += valid ? 1 : 0;
                ++s;
            }
#pragma unroll
        for (int t = kCsSlots; t < 64; ++t) A[t] = __builtin_inff();
        bitonic_sort64(A);
        float m = __builtin_inff();
#pragma unroll
        for (int c = 1; c <= kCsSlots; ++c) {                            // c elements taken from A, n+2-c from B
            // B ascending, index n+1-c  ==  reference's j = c-1:  (n - 2j) * l_t * sg   (:1747)
            const float bv = (c <= n + 1) ? (float)(n + 2 - 2 * c) * l_t * sg : __builtin_inff();
            m = __builtin_fminf(m, __builtin_fmaxf(A[c - 1], bv));
        }
        v1 = u1k - ixw * m / sg;                                         // :1756-1757
        v2 = u2k - iyw * m / sg;
    }
    const float wt = base[N_WT * pe + i];
    float dp = 0.0f, dq = 0.0f;
    int s = 0;
#pragma unroll
    for (int k = -kNlR; k <= kNlR; ++k)
#pragma unroll
        for (int l = -kNlR; l <= kNlR; ++l) {
            if (k == 0 && l == 0) continue;
            if (x + l >= 0 && x + l < nx && y + k >= 0 && y + k < ny) {
                const int ap = i + k * nx + l;
                const int rp = kNlSlots - 1 - s;
                const float w = base[(size_t)(N_W + s) * pe + i];
                dp += w * (base[(size_t)(N_P + s) * pe + i] - base[(size_t)(N_P + rp) * pe + ap]);
                dq += w * (base[(size_t)(N_Q + s) * pe + i] - base[(size_t)(N_Q + rp) * pe + ap]);
            }
            ++s;
        }
    dp /= wt;
    dq /= wt;
    const float n1 = u1k - tau * (dp + (u1k - v1) / theta);              // :1106-1107, note +div
    const float n2 = u2k - tau * (dq + (u2k - v2) / theta);
    base[N_U1 * pe + i] = n1;
    base[N_U2 * pe + i] = n2;
    base[N_UB1 * pe + i] = 2.0f * n1 - u1k;                              // :1774-1777
    base[N_UB2 * pe + i] = 2.0f * n2 - u2k;
}

}  // namespace faldoi
