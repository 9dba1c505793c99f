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
constexpr int kNlBatch = 6;                // non-local slots whose 5x6 loads are issued together
constexpr int kCsBatch = 8;                // CSAD slots whose 2x8 neighbour loads are issued together (16: no faster, more registers)

// Plane indices of one NLTV-CSAD problem
enum NlPlane {
    N_I0 = 0, N_I1, N_I1X, N_I1Y, N_I1W, N_I1WX, N_I1WY, N_GRAD, N_U1, N_U2, N_UB1, N_UB2, N_WT,
    N_P,                                   // 24 planes
    N_Q = N_P + kNlSlots,                  // 24 planes
    N_W = N_Q + kNlSlots,                  // 24 planes
    N_T1 = N_W + kNlSlots,                 // I1wx*u1 and I1wy*u2 at the start of the warp (CSAD terms)
    N_T2,
    // second buffers of what the fused kernel (nltv_fused_kernel.hpp) reads from neighbouring tiles while their owners rewrite it
    N_UBB1, N_UBB2,
    N_PB,                                  // 24 planes
    N_QB = N_PB + kNlSlots,                // 24 planes
    N_RK = N_QB + kNlSlots,                // 12 planes of packed bytes: the CSAD terms' order of this warp (csad_rank_kernel)
    N_COUNT = N_RK + 12
};

// Per-warp: three bicubic warps (border_out = true), grad = I1wx^2 + I1wy^2, ubar = u (:1694-1725)
// data_l1 (NLTV-L1, src/global_faldoi.cpp:1226-1234): the N_I1W plane receives rho_c instead of I1w.
__global__ __launch_bounds__(256) void nl_warp_kernel(float *__restrict__ base, size_t pe, int nx, int ny, int data_l1, int cur)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= nx * ny) return;
    const int y = p / nx, x = p - y * nx;
    const float u1 = base[N_U1 * pe + p], u2 = base[N_U2 * pe + p];
    const BicubicTaps tp = bicubic_taps((float)x + u1, (float)y + u2, nx, ny);
    float w0 = 0.0f, wx = 0.0f, wy = 0.0f;
    if (!tp.clamped) {
        w0 = bicubic_eval_rows(base + N_I1 * pe, tp, nx);
        wx = bicubic_eval_rows(base + N_I1X * pe, tp, nx);
        wy = bicubic_eval_rows(base + N_I1Y * pe, tp, nx);
    }
    base[N_I1W * pe + p] = data_l1 ? rho_const(w0, wx, wy, u1, u2, base[N_I0 * pe + p]) : w0;
    base[N_I1WX * pe + p] = wx;
    base[N_I1WY * pe + p] = wy;
    base[N_GRAD * pe + p] = grad_sq(wx, wy);
    base[N_T1 * pe + p] = wx * u1;                              // the two products inside b (:1716-1717)
    base[N_T2 * pe + p] = wy * u2;
    base[(cur ? N_UBB1 : N_UB1) * pe + p] = u1;                 // ubar = u, into the buffer the next iteration reads
    base[(cur ? N_UBB2 : N_UB2) * pe + p] = u2;
}

// The CSAD terms of the reference (PosNei::b[48], :1704-1719)
//     b_s = (I0[i] - I0[pos_s] - I1w[i] + I1w[pos_s] + I1wx[i]*u1[i] + I1wy[i]*u2[i]) / sqrt(grad[i])
// are per-warp constants that the reference stores (48 floats per pixel) and re-reads every
// iteration.  Here they are recomputed inside the data step from I0, I1w (7x7 neighbourhoods, served
// by L1/L2) and the two warp-start products: 48 planes (86 MB at 1024x436) less traffic per
// iteration and less HBM footprint.  The reference only refreshes b where grad > GRAD_IS_ZERO and
// only reads it there (:1737), so recomputing is equivalent.

// ofnltv_getD (:1127-1174) on ubar: for every valid slot
//   g = w * (ubar_i - ubar_j) / wt;  p <- (p + tau*g) / (1 + tau*sqrt(g*g))     (same for q with ubar2)
__global__ __launch_bounds__(256) void nl_dual_kernel(float *__restrict__ base, size_t pe, int nx, int ny, float tau)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx * ny) return;
    const int y = i / nx, x = i - y * nx;
    const float *ub1 = base + N_UB1 * pe, *ub2 = base + N_UB2 * pe;
    const float a1 = ub1[i], a2 = ub2[i], wt = base[N_WT * pe + i];
    const double rwt = recip_f64(wt);        // wt > 0 (every pixel has valid neighbours): shared by 48 quotients
    // Six slots at a time: their thirty loads are issued together (clamped address for slots outside the image);
    // slots outside the image keep their duals (never written, as in the reference).
#pragma unroll
    for (int s0 = 0; s0 < kNlSlots; s0 += kNlBatch) {
        float w_[kNlBatch], n1[kNlBatch], n2[kNlBatch], p0[kNlBatch], q0[kNlBatch];
        bool ok[kNlBatch];
#pragma unroll
        for (int q = 0; q < kNlBatch; ++q) {
            const int s = s0 + q, c = s < 12 ? s : s + 1;           // position in the 5x5 raster (centre = 12)
            const int k = c / 5 - kNlR, l = c % 5 - kNlR;
            ok[q] = x + l >= 0 && x + l < nx && y + k >= 0 && y + k < ny;
            const int ap = ok[q] ? i + k * nx + l : i;
            w_[q] = base[(size_t)(N_W + s) * pe + i];
            n1[q] = ub1[ap];
            n2[q] = ub2[ap];
            p0[q] = base[(size_t)(N_P + s) * pe + i];
            q0[q] = base[(size_t)(N_Q + s) * pe + i];
        }
#pragma unroll
        for (int q = 0; q < kNlBatch; ++q) {
            if (ok[q]) {
                const int s = s0 + q;
                const float g1 = div_by(w_[q] * (a1 - n1[q]), wt, rwt);
                const float d1 = 1 + tau * sqrt_of_square(g1);
                base[(size_t)(N_P + s) * pe + i] = (p0[q] + tau * g1) / d1;
                const float g2 = div_by(w_[q] * (a2 - n2[q]), wt, rwt);
                const float d2 = 1 + tau * sqrt_of_square(g2);
                base[(size_t)(N_Q + s) * pe + i] = (q0[q] + tau * g2) / d2;
            }
        }
    }
}

// Compare-exchange network for the 48 CSAD terms held in registers: Batcher's odd-even merge sort
// for 64 inputs with every comparator that touches an index >= 48 removed (those inputs would be
// +inf and never move) -- 384 comparators instead of the 672 of a 64-input bitonic network.
// Generated and 0/1-checked by the script quoted in DESIGN.md section 3.4.
struct CxPair { unsigned char a, b; };
__device__ constexpr CxPair kSort48[384] = {
    {0,1}, {2,3}, {0,2}, {1,3}, {1,2}, {4,5}, {6,7}, {4,6}, {5,7}, {5,6}, {0,4}, {2,6},
    {2,4}, {1,5}, {3,7}, {3,5}, {1,2}, {3,4}, {5,6}, {8,9}, {10,11}, {8,10}, {9,11}, {9,10},
    {12,13}, {14,15}, {12,14}, {13,15}, {13,14}, {8,12}, {10,14}, {10,12}, {9,13}, {11,15}, {11,13}, {9,10},
    {11,12}, {13,14}, {0,8}, {4,12}, {4,8}, {2,10}, {6,14}, {6,10}, {2,4}, {6,8}, {10,12}, {1,9},
    {5,13}, {5,9}, {3,11}, {7,15}, {7,11}, {3,5}, {7,9}, {11,13}, {1,2}, {3,4}, {5,6}, {7,8},
    {9,10}, {11,12}, {13,14}, {16,17}, {18,19}, {16,18}, {17,19}, {17,18}, {20,21}, {22,23}, {20,22}, {21,23},
    {21,22}, {16,20}, {18,22}, {18,20}, {17,21}, {19,23}, {19,21}, {17,18}, {19,20}, {21,22}, {24,25}, {26,27},
    {24,26}, {25,27}, {25,26}, {28,29}, {30,31}, {28,30}, {29,31}, {29,30}, {24,28}, {26,30}, {26,28}, {25,29},
    {27,31}, {27,29}, {25,26}, {27,28}, {29,30}, {16,24}, {20,28}, {20,24}, {18,26}, {22,30}, {22,26}, {18,20},
    {22,24}, {26,28}, {17,25}, {21,29}, {21,25}, {19,27}, {23,31}, {23,27}, {19,21}, {23,25}, {27,29}, {17,18},
    {19,20}, {21,22}, {23,24}, {25,26}, {27,28}, {29,30}, {0,16}, {8,24}, {8,16}, {4,20}, {12,28}, {12,20},
    {4,8}, {12,16}, {20,24}, {2,18}, {10,26}, {10,18}, {6,22}, {14,30}, {14,22}, {6,10}, {14,18}, {22,26},
    {2,4}, {6,8}, {10,12}, {14,16}, {18,20}, {22,24}, {26,28}, {1,17}, {9,25}, {9,17}, {5,21}, {13,29},
    {13,21}, {5,9}, {13,17}, {21,25}, {3,19}, {11,27}, {11,19}, {7,23}, {15,31}, {15,23}, {7,11}, {15,19},
    {23,27}, {3,5}, {7,9}, {11,13}, {15,17}, {19,21}, {23,25}, {27,29}, {1,2}, {3,4}, {5,6}, {7,8},
    {9,10}, {11,12}, {13,14}, {15,16}, {17,18}, {19,20}, {21,22}, {23,24}, {25,26}, {27,28}, {29,30}, {32,33},
    {34,35}, {32,34}, {33,35}, {33,34}, {36,37}, {38,39}, {36,38}, {37,39}, {37,38}, {32,36}, {34,38}, {34,36},
    {33,37}, {35,39}, {35,37}, {33,34}, {35,36}, {37,38}, {40,41}, {42,43}, {40,42}, {41,43}, {41,42}, {44,45},
    {46,47}, {44,46}, {45,47}, {45,46}, {40,44}, {42,46}, {42,44}, {41,45}, {43,47}, {43,45}, {41,42}, {43,44},
    {45,46}, {32,40}, {36,44}, {36,40}, {34,42}, {38,46}, {38,42}, {34,36}, {38,40}, {42,44}, {33,41}, {37,45},
    {37,41}, {35,43}, {39,47}, {39,43}, {35,37}, {39,41}, {43,45}, {33,34}, {35,36}, {37,38}, {39,40}, {41,42},
    {43,44}, {45,46}, {36,40}, {38,42}, {34,36}, {38,40}, {42,44}, {37,41}, {39,43}, {35,37}, {39,41}, {43,45},
    {33,34}, {35,36}, {37,38}, {39,40}, {41,42}, {43,44}, {45,46}, {0,32}, {16,32}, {8,40}, {24,40}, {8,16},
    {24,32}, {4,36}, {20,36}, {12,44}, {28,44}, {12,20}, {28,36}, {4,8}, {12,16}, {20,24}, {28,32}, {36,40},
    {2,34}, {18,34}, {10,42}, {26,42}, {10,18}, {26,34}, {6,38}, {22,38}, {14,46}, {30,46}, {14,22}, {30,38},
    {6,10}, {14,18}, {22,26}, {30,34}, {38,42}, {2,4}, {6,8}, {10,12}, {14,16}, {18,20}, {22,24}, {26,28},
    {30,32}, {34,36}, {38,40}, {42,44}, {1,33}, {17,33}, {9,41}, {25,41}, {9,17}, {25,33}, {5,37}, {21,37},
    {13,45}, {29,45}, {13,21}, {29,37}, {5,9}, {13,17}, {21,25}, {29,33}, {37,41}, {3,35}, {19,35}, {11,43},
    {27,43}, {11,19}, {27,35}, {7,39}, {23,39}, {15,47}, {31,47}, {15,23}, {31,39}, {7,11}, {15,19}, {23,27},
    {31,35}, {39,43}, {3,5}, {7,9}, {11,13}, {15,17}, {19,21}, {23,25}, {27,29}, {31,33}, {35,37}, {39,41},
    {43,45}, {1,2}, {3,4}, {5,6}, {7,8}, {9,10}, {11,12}, {13,14}, {15,16}, {17,18}, {19,20}, {21,22},
    {23,24}, {25,26}, {27,28}, {29,30}, {31,32}, {33,34}, {35,36}, {37,38}, {39,40}, {41,42}, {43,44}, {45,46},
};
__device__ __forceinline__ void cswap(float &a, float &b)
{
    const float lo = __builtin_fminf(a, b);   // v_min_f32 / v_max_f32; -0 orders before +0, which
    const float hi = __builtin_fmaxf(a, b);   // std::sort leaves unspecified
    a = lo;
    b = hi;
}
__device__ __forceinline__ void sort48(float (&v)[kCsSlots])
{
#pragma unroll
    for (int k = 0; k < 384; ++k) cswap(v[kSort48[k].a], v[kSort48[k].b]);
}

// The CSAD selection shared by NLTV-CSAD (:1733-1758) and TV-CSAD (:1547-1571): returns the element of index
// n+1 (0-based; `it/2+1` with it = 2n+1, sic) of the ascending sort of
//   A = { -(b_s - r) : valid s }  united with  B = { (n-2j)*l_t*sg : j = 0..n },
// b_s = (I0[i] - I0[pos_s] - I1w[i] + I1w[pos_s] + t1 + t2) / sg recomputed from the 7x7 neighbourhoods.
// A is sorted in registers (invalid slots = +inf); B is an arithmetic progression, so the order statistic
// is  min over c of max(A_sorted[c-1], B_ascending[n+1-c])  -- the classic two-sorted-arrays split.
__device__ __forceinline__ float csad_select(const float *I0, const float *I1w, int i, int x, int y,
                                             int nx, int ny, float t1, float t2, float r, float sg, float l_t)
{
    const float i0 = I0[i], i1w = I1w[i];
    const double rsg = recip_f64(sg);                       // sg >= 1e-4: 48 exact quotients share it
    float A[kCsSlots];
    int n = 0;
    // slot s <-> offset (k, l): rows -3..3, columns -3..3, centre skipped.  Eight slots at a time: their sixteen
    // loads are issued together, then the eight quotients (the compiler otherwise waits for every pair of loads).
#pragma unroll
    for (int s0 = 0; s0 < kCsSlots; s0 += kCsBatch) {
        float a0[kCsBatch], a1[kCsBatch];
        bool ok[kCsBatch];
#pragma unroll
        for (int q = 0; q < kCsBatch; ++q) {
            const int s = s0 + q, c = s < 24 ? s : s + 1;           // position in the 7x7 raster (centre = 24)
            const int k = c / 7 - kCsR, l = c % 7 - kCsR;
            ok[q] = x + l >= 0 && x + l < nx && y + k >= 0 && y + k < ny;
            const int pos = ok[q] ? i + k * nx + l : i;
            a0[q] = I0[pos];
            a1[q] = I1w[pos];
        }
#pragma unroll
        for (int q = 0; q < kCsBatch; ++q) {
            const float bs = div_by(i0 - a0[q] - i1w + a1[q] + t1 + t2, sg, rsg);       // :1716-1717 / :1529-1530
            A[s0 + q] = ok[q] ? -(bs - r) : __builtin_inff();       // :1741-1742
            n += ok[q] ? 1 : 0;
        }
    }
    sort48(A);
    float m = __builtin_inff();
#pragma unroll
    for (int c = 1; c <= kCsSlots; ++c) {                            // c elements taken from A, n+2-c from B
        // B ascending, index n+1-c  ==  reference's j = c-1:  (n - 2j) * l_t * sg   (:1747)
        const float bv = (c <= n + 1) ? (float)(n + 2 - 2 * c) * l_t * sg : __builtin_inff();
        m = __builtin_fminf(m, __builtin_fmaxf(A[c - 1], bv));
    }
    return m;
}

// ---------------------------------------------------------------------------------------------
// The same selection WITHOUT a sort per iteration (round 4).
//
// Within one warp the 48 CSAD terms b_s are constants (I0, I1w and the two warp-start products do not change), and every
// iteration sorts A_s = -(b_s - r) with an r common to the 48 slots: x -> -fl(x - r) is monotone, so the ORDER of A is the
// order of b (descending) for the whole warp.  csad_rank_kernel sorts once per warp and stores, per pixel, which slot holds
// rank 0, 1, ... (48 bytes).  With the order known the selection is a search, not a sort: sorted A is non-decreasing,
// C_rho = (n - 2 rho) l_t sg is strictly decreasing, so T_rho = max(A_(rho), C_rho) falls (= C_rho) while A_(rho) < C_rho and
// rises (= A_(rho)) afterwards, and  min_c max(A[c-1], B[n+1-c]) = min_rho T_rho  sits at the crossing: a lower-bound search
// over rho with steps 32, 16, ..., 1 probes six ranks; the largest probed rank with A < C is rho* - 1 and the smallest probed
// one without is rho*, every other probed T is no smaller, so the minimum over the six probed T IS the minimum over all 48.
// Six CSAD terms are evaluated per pixel and iteration instead of 48, and no network (1 500 -> ~200 vector instructions).
// "<" and max / min are taken in the hardware's total order (v_max_f32 / v_min_f32: -0 before +0) exactly as the sorting
// network did, so the result has the same bits as csad_select for every non-NaN input (a probe counts as "A < C" when
// max(A, C) is not A itself) -- checked on the device against the network form (FALDOI_CSAD_SORT=1 keeps it selectable on the
// streaming kernels) and against the CPU checker, which sorts like the reference.
//
// Table layout (12 dwords per pixel, plane w = dword w): the search visits rank (2p+1) step - 1 at the level with that step,
// p = the path taken so far; a level's candidates lie together so that one v_perm_b32 picks the byte:
//   dwords 0-1: level 4 (ranks 3, 11, .., 43), then level 2 (15, 47)      dword 2: level 3 (7, 23, 39), then level 1 (31)
//   dwords 3-5: level 5 (ranks 1, 5, .., 45)                               dwords 6-11: level 6 (ranks 0, 2, .., 46)
// A byte is the slot's window position (k+3)*8 + (l+3).  Ranks >= n (slots outside the image) hold 0; a probe beyond n is
// not evaluated ("guard"), its address is still a legal one.
// ---------------------------------------------------------------------------------------------
constexpr int kCsRankWords = 12;
__device__ constexpr unsigned char kCsRankOfByte[4 * kCsRankWords] = {
    3, 11, 19, 27, 35, 43, 15, 47,   7, 23, 39, 31,
    1, 5, 9, 13, 17, 21, 25, 29, 33, 37, 41, 45,
    0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46,
};
__device__ __forceinline__ int csad_order_key(float v)      // integer order == total order of the floats (-0 < +0)
{
    const int b = __builtin_bit_cast(int, v);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ int csad_valid_count(int x, int y, int nx, int ny)      // neighbours of the 7x7 window inside the image
{
    const int cx = (x < kCsR ? x : kCsR) + (nx - 1 - x < kCsR ? nx - 1 - x : kCsR) + 1;
    const int cy = (y < kCsR ? y : kCsR) + (ny - 1 - y < kCsR ? ny - 1 - y : kCsR) + 1;
    return cx * cy - 1;
}

// Once per warp, after the warp kernel (it needs the whole I1w plane): order of the numerators
//   i0 - I0[pos_s] - i1w + I1w[pos_s] + t1 + t2     (:1716-1717 / :1529-1530 before the division by sg > 0: same order as b)
// descending, by the 48-input network on (key, position byte) pairs; equal keys give equal A, so their order is immaterial.
__global__ __launch_bounds__(256) void csad_rank_kernel(const float *__restrict__ I0, const float *__restrict__ I1w,
                                                        const float *__restrict__ T1, const float *__restrict__ T2,
                                                        unsigned *__restrict__ rk, size_t pe, int nx, int ny)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nx * ny) return;
    const int y = i / nx, x = i - y * nx;
    const float i0 = I0[i], i1w = I1w[i], t1 = T1[i], t2 = T2[i];
    int key[kCsSlots];
    unsigned pay[kCsSlots];
#pragma unroll
    for (int s0 = 0; s0 < kCsSlots; s0 += kCsBatch) {
        float a0[kCsBatch], a1[kCsBatch];
        bool ok[kCsBatch];
#pragma unroll
        for (int q = 0; q < kCsBatch; ++q) {
            const int s = s0 + q, c = s < 24 ? s : s + 1;
            const int k = c / 7 - kCsR, l = c % 7 - kCsR;
            ok[q] = x + l >= 0 && x + l < nx && y + k >= 0 && y + k < ny;
            const int pos = ok[q] ? i + k * nx + l : i;
            a0[q] = I0[pos];
            a1[q] = I1w[pos];
        }
#pragma unroll
        for (int q = 0; q < kCsBatch; ++q) {
            const int s = s0 + q, c = s < 24 ? s : s + 1;
            const int k = c / 7 - kCsR, l = c % 7 - kCsR;
            const float num = i0 - a0[q] - i1w + a1[q] + t1 + t2;
            key[s] = ok[q] ? csad_order_key(num) : (int)0x80000000;            // below every non-NaN key: sorts last
            pay[s] = ok[q] ? (unsigned)((k + kCsR) * 8 + (l + kCsR)) : 0u;
        }
    }
#pragma unroll
    for (int c = 0; c < 384; ++c) {                                           // descending: the larger key to the lower index
        const int a = kSort48[c].a, b = kSort48[c].b;
        const bool sw = key[a] < key[b];
        const int ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];
        const unsigned pa = sw ? pay[b] : pay[a], pb = sw ? pay[a] : pay[b];
        key[a] = ka; key[b] = kb; pay[a] = pa; pay[b] = pb;
    }
#pragma unroll
    for (int w = 0; w < kCsRankWords; ++w) {
        const unsigned v = pay[kCsRankOfByte[4 * w]] | (pay[kCsRankOfByte[4 * w + 1]] << 8) | (pay[kCsRankOfByte[4 * w + 2]] << 16) |
                           (pay[kCsRankOfByte[4 * w + 3]] << 24);
        rk[(size_t)w * pe + i] = v;
    }
}

// The byte of the level's candidate p (see the layout above).  One past a level's last candidate (the path of a pixel whose
// every term lies below the progression) still returns a byte of the table.
__device__ __forceinline__ unsigned csad_perm(unsigned hi, unsigned lo, unsigned sel)      // byte sel (0..7) of {hi, lo}
{
    return __builtin_amdgcn_perm(hi, lo, sel | 0x0c0c0c00u);
}
template <int LEVEL>
__device__ __forceinline__ unsigned csad_rank_byte(const unsigned (&d)[kCsRankWords], unsigned p)
{
    if (LEVEL == 1) return d[2] >> 24;
    if (LEVEL == 2) return csad_perm(d[1], d[0], p + 6u);
    if (LEVEL == 3) return csad_perm(d[2], d[2], p);
    if (LEVEL == 4) return csad_perm(d[1], d[0], p);
    if (LEVEL == 5) {
        const unsigned sel = p & 7u;
        const unsigned a = csad_perm(d[4], d[3], sel), b = csad_perm(d[3], d[5], sel);
        return p < 8u ? a : b;
    }
    const unsigned sel = p & 7u;
    const unsigned a = csad_perm(d[7], d[6], sel), b = csad_perm(d[9], d[8], sel), c = csad_perm(d[11], d[10], sel);
    return p < 8u ? a : (p < 16u ? b : c);
}

// fetch(e, live, a0, a1): I0 and I1w of the neighbour at window position byte e (live = false: any legal address, value unused).
// ALL48: every pixel this instantiation sees has its 48 neighbours (n == 48, guards are constants of the level).
template <bool ALL48, class Fetch>
__device__ __forceinline__ float csad_select_ranked(const unsigned (&d)[kCsRankWords], int n, Fetch fetch, float i0, float i1w,
                                                    float t1, float t2, float r, float sg, double rsg, float l_t)
{
    float m = __builtin_inff();
    unsigned p = 0u;
#define FALDOI_CSAD_LEVEL(LEVEL, STEP)                                                                                         \
    {                                                                                                                          \
        /* rank probed: rho = (2p+1) STEP - 1; it exists iff (2p+1) STEP <= n */                                               \
        const bool live = ALL48 ? (STEP >= 16 || p <= (unsigned)((48 - STEP) / (2 * STEP)))                                    \
                                : (int)((2u * p + 1u) * (unsigned)STEP) <= n;                                                 \
        const unsigned e = csad_rank_byte<LEVEL>(d, p);                                                                        \
        float a0, a1;                                                                                                          \
        fetch(e, live, a0, a1);                                                                                                \
        const float bs = div_by(i0 - a0 - i1w + a1 + t1 + t2, sg, rsg);                  /* :1716-1717 / :1529-1530 */         \
        const float A = live ? -(bs - r) : __builtin_inff();                              /* :1741-1742 */                     \
        const int coef = (n + 2 - 2 * STEP) - 4 * STEP * (int)p;                          /* n - 2 rho */                      \
        const float C = (float)coef * l_t * sg;                                           /* :1747 */                          \
        const float T = __builtin_fmaxf(A, C);                                                                                 \
        m = __builtin_fminf(m, T);                                                                                             \
        p = 2u * p + (__builtin_bit_cast(unsigned, T) != __builtin_bit_cast(unsigned, A) ? 1u : 0u);                           \
    }
    FALDOI_CSAD_LEVEL(1, 32)
    FALDOI_CSAD_LEVEL(2, 16)
    FALDOI_CSAD_LEVEL(3, 8)
    FALDOI_CSAD_LEVEL(4, 4)
    FALDOI_CSAD_LEVEL(5, 2)
    FALDOI_CSAD_LEVEL(6, 1)
#undef FALDOI_CSAD_LEVEL
    return m;
}

// Neighbours from the planes themselves (streaming kernels)
struct CsadFetchGlobal {
    const float *I0, *I1w;
    int i, nx;
    __device__ __forceinline__ void operator()(unsigned e, bool live, float &a0, float &a1) const
    {
        const int pos = live ? i + ((int)(e >> 3) - kCsR) * nx + ((int)(e & 7u) - kCsR) : i;
        a0 = I0[pos];
        a1 = I1w[pos];
    }
};
__device__ __forceinline__ float csad_select_by_ranks(const float *I0, const float *I1w, const unsigned *rk, size_t pe, int i, int x, int y,
                                                      int nx, int ny, float t1, float t2, float r, float sg, float l_t)
{
    unsigned d[kCsRankWords];
#pragma unroll
    for (int w = 0; w < kCsRankWords; ++w) d[w] = rk[(size_t)w * pe + i];
    const double rsg = recip_f64(sg);
    const int n = csad_valid_count(x, y, nx, ny);
    return csad_select_ranked<false>(d, n, CsadFetchGlobal{ I0, I1w, i, nx }, I0[i], I1w[i], t1, t2, r, sg, rsg, l_t);
}

// Data step (:1733-1758) + non-local divergence (:1056-1079) + ofnltv_getP (:1090-1120) + relaxation.
// The data step needs the element of index n+1 (0-based; `it/2+1` with it = 2n+1, sic) of the
// ascending sort of  A = { -(b_s - r) : valid s }  united with  B = { (n-2j)*l_t*sg : j = 0..n }.
// A is sorted in registers (invalid slots = +inf); B is an arithmetic progression, so the order statistic
// is  min over i of max(A_sorted[i-1], B_ascending[n+1-i])  -- the classic two-sorted-arrays split.
// DATA_L1 = true: the data step is TV-L1's thresholding (nltvl1_PD, :1246-1272) on rho_c.
// RANKED: the selection by the warp's rank table (csad_select_by_ranks); false: the sorting network (FALDOI_CSAD_SORT=1, a device-side
// cross-check of the former).
template <bool DATA_L1, bool RANKED = true>
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
        const float m = RANKED ? csad_select_by_ranks(base + N_I0 * pe, base + N_I1W * pe, reinterpret_cast<const unsigned *>(base + N_RK * pe),
                                                      pe, i, x, y, nx, ny, base[N_T1 * pe + i], base[N_T2 * pe + i], r, sg, l_t)
                               : csad_select(base + N_I0 * pe, base + N_I1W * pe, i, x, y, nx, ny, base[N_T1 * pe + i], base[N_T2 * pe + i],
                                             r, sg, l_t);
        v1 = u1k - ixw * m / sg;                                         // :1756-1757
        v2 = u2k - iyw * m / sg;
    }
    const float wt = base[N_WT * pe + i];
    float dp = 0.0f, dq = 0.0f;
    // non_local_divergence (:1056-1079): slot s <-> offset (k, l) in the 5x5 raster, centre skipped; the mirror of slot s is
    // slot 23 - s at the opposite offset.  A dual plane is needed twice -- p_s[i] for slot s and p_s[i - off_s] for slot 23 - s --
    // so the two slots of a mirror pair are loaded together (three pairs = 30 loads in flight): the second touch of a plane
    // finds its lines in L1 / L2 instead of streaming the 48 planes through twice.  The terms are independent; the SUM keeps
    // the reference's slot order 0..23.
    float tp[kNlSlots], tq[kNlSlots];
    bool oks[kNlSlots];
#pragma unroll
    for (int s0 = 0; s0 < kNlSlots / 2; s0 += kNlBatch / 2) {
        float w_[kNlBatch], pa[kNlBatch], pb[kNlBatch], qa[kNlBatch], qb[kNlBatch];
        int sl[kNlBatch];
#pragma unroll
        for (int q = 0; q < kNlBatch; ++q) {
            const int s = (q & 1) ? kNlSlots - 1 - (s0 + q / 2) : s0 + q / 2, c = s < 12 ? s : s + 1;   // s, its mirror, s+1, ...
            const int k = c / 5 - kNlR, l = c % 5 - kNlR;
            sl[q] = s;
            oks[s] = x + l >= 0 && x + l < nx && y + k >= 0 && y + k < ny;
            const int ap = oks[s] ? i + k * nx + l : i;
            const int rp = kNlSlots - 1 - s;
            w_[q] = base[(size_t)(N_W + s) * pe + i];
            pa[q] = base[(size_t)(N_P + s) * pe + i];
            pb[q] = base[(size_t)(N_P + rp) * pe + ap];
            qa[q] = base[(size_t)(N_Q + s) * pe + i];
            qb[q] = base[(size_t)(N_Q + rp) * pe + ap];
        }
#pragma unroll
        for (int q = 0; q < kNlBatch; ++q) {
            tp[sl[q]] = w_[q] * (pa[q] - pb[q]);
            tq[sl[q]] = w_[q] * (qa[q] - qb[q]);
        }
    }
#pragma unroll
    for (int s = 0; s < kNlSlots; ++s) {
        const float np = dp + tp[s], nq = dq + tq[s];
        dp = oks[s] ? np : dp;
        dq = oks[s] ? nq : dq;
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

// ---------------------------------------------------------------------------------------------
// TV-CSAD (tvcsad_PD, src/global_faldoi.cpp:1449-1636; global_faldoi -m 4 / 5): the CSAD data term over
// sg = hypot(Ix^2 + Iy^2, 0.01) with a TV regulariser normalised per flow component (tvcsad_getD :1424).
// Streams like the solvers above: one launch for the dual step, one for data + divergence + primal step.
// ---------------------------------------------------------------------------------------------
enum TcPlane {
    C_I0 = 0, C_I1, C_I1X, C_I1Y, C_I1W, C_I1WX, C_I1WY, C_GRAD, C_T1, C_T2, C_U1, C_U2, C_UB1, C_UB2,
    C_X11, C_X12, C_X21, C_X22, C_ERR,
    C_RK,                                  // 12 planes: the CSAD terms' order of this warp (csad_rank_kernel)
    C_COUNT = C_RK + 12
};

// Per warp (:1509-1538): warps, the two products inside b, ubar = u.  C_GRAD receives Ix^2 + Iy^2; the host
// replaces it by (float)hypot(that, 0.01) with the C library's double hypot, as the reference computes it.
__global__ __launch_bounds__(256) void tc_warp_kernel(float *__restrict__ base, size_t pe, int nx, int ny)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= nx * ny) return;
    const int y = p / nx, x = p - y * nx;
    const float u1 = base[C_U1 * pe + p], u2 = base[C_U2 * pe + p];
    const BicubicTaps tp = bicubic_taps((float)x + u1, (float)y + u2, nx, ny);
    float w0 = 0.0f, wx = 0.0f, wy = 0.0f;
    if (!tp.clamped) {
        w0 = bicubic_eval_rows(base + C_I1 * pe, tp, nx);
        wx = bicubic_eval_rows(base + C_I1X * pe, tp, nx);
        wy = bicubic_eval_rows(base + C_I1Y * pe, tp, nx);
    }
    base[C_I1W * pe + p] = w0;
    base[C_I1WX * pe + p] = wx;
    base[C_I1WY * pe + p] = wy;
    base[C_GRAD * pe + p] = grad_sq(wx, wy);
    base[C_T1 * pe + p] = wx * u1;
    base[C_T2 * pe + p] = wy * u2;
    base[C_UB1 * pe + p] = u1;
    base[C_UB2 * pe + p] = u2;
}

// Loop control of one warp, on the device: the host queues several iterations at a time; once the stop test has
// been decided (or cannot be decided without the serial sum) the kernels that are already queued do nothing.
struct TcCtl {
    int halt;            // 0: keep iterating; 1: stopped (test decided, or iteration cap); 2: the host must settle the test
    int iters;           // iterations carried out in this warp
    int pad[2];
    double mean;         // mean |du|^2 of the last iteration (order-independent double sum)
};

// hypotf as the C library evaluates it for finite arguments: sqrt(x*x + y*y) in double, rounded to float
__device__ __forceinline__ float hypotf_like_libm(float a, float b)
{
    if (__builtin_isinf(a) || __builtin_isinf(b)) return __builtin_inff();
    return (float)__builtin_sqrt((double)a * (double)a + (double)b * (double)b);
}

// forward_gradient of ubar (src/utils.cpp:285) + tvcsad_getD (:1424-1441)
__global__ __launch_bounds__(256) void tc_dual_kernel(float *__restrict__ base, size_t pe, int nx, int ny, float tau,
                                                      const TcCtl *__restrict__ ctl)
{
    if (ctl->halt) return;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= nx * ny) return;
    const int y = p / nx, x = p - y * nx;
    const bool lc = x == nx - 1, lr = y == ny - 1;
    const float *ub1 = base + C_UB1 * pe, *ub2 = base + C_UB2 * pe;
    const float a = ub1[p], b = ub2[p];
    const float ar = ub1[lc ? p : p + 1], ad = ub1[lr ? p : p + nx], br = ub2[lc ? p : p + 1], bd = ub2[lr ? p : p + nx];
    const float u1x = fwd_diff(ar, a, lc), u1y = fwd_diff(ad, a, lr), u2x = fwd_diff(br, b, lc), u2y = fwd_diff(bd, b, lr);
    float x11 = base[C_X11 * pe + p], x12 = base[C_X12 * pe + p], x21 = base[C_X21 * pe + p], x22 = base[C_X22 * pe + p];
    float n1 = hypotf_like_libm(x11, x12), n2 = hypotf_like_libm(x21, x22);
    n1 = (1 > n1) ? 1 : n1;
    n2 = (1 > n2) ? 1 : n2;
    base[C_X11 * pe + p] = (x11 + tau * u1x) / n1;
    base[C_X12 * pe + p] = (x12 + tau * u1y) / n1;
    base[C_X21 * pe + p] = (x21 + tau * u2x) / n2;
    base[C_X22 * pe + p] = (x22 + tau * u2y) / n2;
}

// Data step (:1547-1571) + divergence (src/utils.cpp:239) + tvcsad_getP (:1386-1414) + relaxation (:1593-1597).
// The squared update of every pixel goes to C_ERR (the reference's stop test sums them in pixel order) and
// a per-block double sum to part[] (order-independent to ~1e-13: enough to decide the test whenever the mean
// is not within the rounding slack of a serial fp32 sum around tol^2; the host settles the rest exactly).
template <bool RANKED>
__global__ __launch_bounds__(256) void tc_primal_kernel(float *__restrict__ base, size_t pe, int nx, int ny, float l_t,
                                                        float theta, float tau, double *__restrict__ part,
                                                        const TcCtl *__restrict__ ctl)
{
    __shared__ double red[256];
    if (ctl->halt) return;                                              // uniform over the grid
    const int i = blockIdx.x * 256 + threadIdx.x;
    float tt = 0.0f;
    if (i < nx * ny) {
        const int y = i / nx, x = i - y * nx;
        const float u1k = base[C_U1 * pe + i], u2k = base[C_U2 * pe + i];
        const float sg = base[C_GRAD * pe + i];
        const float ixw = base[C_I1WX * pe + i], iyw = base[C_I1WY * pe + i];
        const float r = (ixw * u1k + iyw * u2k) / sg;
        const float m = RANKED ? csad_select_by_ranks(base + C_I0 * pe, base + C_I1W * pe, reinterpret_cast<const unsigned *>(base + C_RK * pe),
                                                      pe, i, x, y, nx, ny, base[C_T1 * pe + i], base[C_T2 * pe + i], r, sg, l_t)
                               : csad_select(base + C_I0 * pe, base + C_I1W * pe, i, x, y, nx, ny, base[C_T1 * pe + i], base[C_T2 * pe + i],
                                             r, sg, l_t);
        const float v1 = u1k - ixw * m / sg;                             // :1568-1569
        const float v2 = u2k - iyw * m / sg;
        const PixClass pc = classify(x, y, nx, ny);
        const float *X11 = base + C_X11 * pe, *X12 = base + C_X12 * pe, *X21 = base + C_X21 * pe, *X22 = base + C_X22 * pe;
        const int pl = x > 0 ? i - 1 : i, pu = y > 0 ? i - nx : i;
        const float d1 = div_sample(X11[i], X11[pl], X12[i], X12[pu], pc);
        const float d2 = div_sample(X21[i], X21[pl], X22[i], X22[pu], pc);
        const float n1 = u1k - tau * (-d1 + (u1k - v1) / theta);
        const float n2 = u2k - tau * (-d2 + (u2k - v2) / theta);
        tt = (n1 - u1k) * (n1 - u1k) + (n2 - u2k) * (n2 - u2k);
        base[C_U1 * pe + i] = n1;
        base[C_U2 * pe + i] = n2;
        base[C_UB1 * pe + i] = 2.0f * n1 - u1k;
        base[C_UB2 * pe + i] = 2.0f * n2 - u2k;
        base[C_ERR * pe + i] = tt;
    }
    red[threadIdx.x] = (double)tt;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// Fixed-order double sum of the block partials, then the stop test of :1542 where it can be decided without knowing
// the order of the reference's fp32 accumulation (`slack`: the relative error bound of a serial fp32 sum of n terms).
__global__ __launch_bounds__(256) void tc_sum_kernel(const double *__restrict__ part, int nparts, TcCtl *__restrict__ ctl, double n,
                                                     double tol2, double slack, int max_inner)
{
    __shared__ double red[256];
    if (ctl->halt) return;
    double acc = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) acc += part[k];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = red[0] / n;
        const int it = ctl->iters + 1;
        ctl->iters = it;
        ctl->mean = mean;
        if (mean * (1.0 - slack) > tol2) ctl->halt = (it >= max_inner) ? 1 : 0;          // err_D > tol^2 whatever the order
        else if (mean * (1.0 + slack) < tol2) ctl->halt = 1;                             // err_D <= tol^2 whatever the order
        else ctl->halt = 2;
    }
}

}  // namespace faldoi
