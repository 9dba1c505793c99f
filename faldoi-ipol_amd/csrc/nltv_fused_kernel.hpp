// nltv_fused_kernel.hpp -- one launch per inner iteration of the NLTV solvers (NLTV-CSAD, BASELINE configs[4]; NLTV-L1).
//
// The two streaming kernels of nltv_kernels.hpp write the 48 dual planes and read them straight back (the divergence needs, for
// every slot s of a pixel, the MIRROR dual p[23-s] of the neighbour the slot points at).  Here a workgroup owns a tile of
// 64 x TY pixels and keeps the new duals of the tile in LDS between the two steps:
//
//   stage      ubar (2-pixel ring) and, for the CSAD data term, I0 / I1w (3-pixel ring) into LDS
//   data       CSAD selection (:1733-1758; neighbourhoods from LDS) or TV-L1 thresholding: v from u
//   dual, p    every thread updates the 24 duals p[s] of its pixel (ofnltv_getD, src/global_faldoi.cpp:1127): new value to HBM,
//              to LDS and kept in registers; the duals of ring pixels whose slot points INTO the tile ("crossing edges",
//              ~10 % of the tile's own) are recomputed from the neighbour tile's old values -- nothing else of the ring is needed
//   div, p     non_local_divergence (:1056): the mirror duals come from LDS at compile-time offsets
//   dual, q / div, q    the same for the second flow component (the LDS planes are reused)
//   primal     ofnltv_getP (:1090), relaxation
//
// A neighbour tile may already have stored its new duals / ubar when this one reads the ring, so both are double buffered in
// HBM (read `cur`, write `1 - cur`; the host flips per iteration).  The per-pixel arithmetic is the two-kernel path's, operation
// for operation (shared device functions), so results are bit-identical to it and to the CPU checker (tests, fuzz family nltv).
//
// LDS: 24 x (TY+4) x 68 floats of duals + the staged planes = 138 KB at TY = 14: one workgroup of 64 x TY threads per CU.
#pragma once
#include "nltv_kernels.hpp"

#pragma clang fp contract(off)

// FALDOI_NFWHATIF (timing-only builds, wrong results; tools/nl_whatif.sh): 1 no data step | 2 no crossing edges | 4 trivial dual
// arithmetic | 8 no sorting network | 16 no HBM stores of the duals | 32 no quotient in the CSAD terms
#ifndef FALDOI_NFWHATIF
#define FALDOI_NFWHATIF 0
#endif


namespace faldoi {

constexpr int kNfWhatIf = FALDOI_NFWHATIF;
constexpr int kNfW = 64;                      // tile width: one wavefront per tile row
constexpr int kNfLW = kNfW + 2 * kNlR;        // 68: LDS row of a dual plane / of ubar (2-pixel ring)
constexpr int kNfIW = kNfW + 2 * kCsR;        // 70: LDS row of the staged I0 / I1w (3-pixel ring)

template <int TY> struct NfLds {
    float pq[kNlSlots][TY + 2 * kNlR][kNfLW];
    float ub[2][TY + 2 * kNlR][kNfLW];
    float im[2][TY + 2 * kCsR][kNfIW];
};

struct NfArgs {
    float *base;
    size_t pe;
    int nx, ny, ntx, nty;
    int cur;                                  // which buffer of the duals / ubar holds the state this iteration reads
    float tau, l_t, theta;
    double rtheta;                            // 1 / theta in double (host), for the exact quotients by theta
    float kl[kCsSlots];                       // (float)(50 - 2c) * l_t, c = 1..48: the arithmetic progression of the CSAD selection when all
                                              // 48 neighbours exist (interior tiles); the same fp32 product the kernel would form
};

// Plane accesses go through ONE buffer resource over the whole slab: the per-lane byte offset of a pixel sits in one VGPR and the
// plane's offset (uniform) in an SGPR, instead of a 64-bit address per access (host: the slab is below 4 GB, else the two-kernel
// path runs).
typedef __amdgpu_buffer_rsrc_t NfRsrc;
__device__ __forceinline__ float nf_ld(NfRsrc r, unsigned plane_bytes, unsigned voff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, plane_bytes, 0));
}
__device__ __forceinline__ void nf_st(NfRsrc r, unsigned plane_bytes, unsigned voff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, plane_bytes, 0);
}

// ofnltv_getD for one slot (src/global_faldoi.cpp:1145-1165): g = w (ubar_i - ubar_j) / wt;  p <- (p + tau g) / (1 + tau sqrt(g g))
// The reference's |g| is sqrt(g*g).  sqrt(RN(g*g)) == |g| for 2^-60 <= |g| <= 2^60 (faldoi_device.hpp, device-checked); below that
// range the two may differ (g*g underflows) but both are <= 2^-60, so for tau < 2^35 (host-checked) tau * either is <= 2^-25 and
// 1 + it rounds to 1 in both forms: the divisor is the same.  Only |g| > 2^60 (or NaN) needs the IEEE root: one compare per
// update, and a wave-uniform branch that is never taken on real data.
__device__ __forceinline__ float nl_dual_update(float p0, float w, float a, float n, float wt, double rwt, float tau)
{
    if (kNfWhatIf & 4) return p0 + w * (a - n) * wt;
    const float g = div_by(w * (a - n), wt, rwt);
    float ag = __builtin_fabsf(g);
    if (__builtin_amdgcn_ballot_w64(!(ag <= 0x1p60f)) != 0ull) {
        asm volatile("" ::: "memory");                                 // keeps this a branch (see sqrt_of_square)
        ag = __builtin_sqrtf(g * g);
    }
    const float d = 1 + tau * ag;
    return (p0 + tau * g) / d;
}
// The same for N slots at once with ONE range test (on the largest |g|): a wave alone on its SIMD pays every branch with idle issue
// slots, and the test is never true on real data.
template <int N>
__device__ __forceinline__ void nl_dual_update_n(float (&p)[N], const float (&w)[N], float a, const float (&n)[N], float wt, double rwt,
                                                 float tau)
{
    float g[N], ag[N];
    unsigned top = 0u;                                                // largest |g| as an integer: NaN and infinity order above every finite value
#pragma unroll
    for (int q = 0; q < N; ++q) {
        g[q] = (kNfWhatIf & 4) ? w[q] * (a - n[q]) * wt : div_by(w[q] * (a - n[q]), wt, rwt);
        const unsigned ab = __float_as_uint(g[q]) & 0x7fffffffu;
        ag[q] = __uint_as_float(ab);
        top = ab > top ? ab : top;
    }
    if (__builtin_amdgcn_ballot_w64(top > 0x5d800000u /* 2^60 */) != 0ull) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < N; ++q) ag[q] = __builtin_sqrtf(g[q] * g[q]);
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const float d = 1 + tau * ag[q];
        p[q] = (kNfWhatIf & 4) ? p[q] + g[q] : (p[q] + tau * g[q]) / d;
    }
}

// Offsets of slot s in the 5x5 raster (rows first, centre skipped)
__device__ __forceinline__ constexpr int nl_slot_k(int s) { return (s < 12 ? s : s + 1) / 5 - kNlR; }
__device__ __forceinline__ constexpr int nl_slot_l(int s) { return (s < 12 ? s : s + 1) % 5 - kNlR; }

// One crossing edge: ring pixel (jx, jy), its slot s, whose neighbour (jx + l, jy + k) lies inside the tile.
struct NfCross {
    int jx, jy, s, k, l;
    bool need;
};
template <int TY>
__device__ __forceinline__ NfCross nf_cross_item(int e, int X0, int Y0, int nx, int ny)
{
    constexpr int nRow = 15 * kNfLW, nCol = 15 * TY;               // per side: (10 slots of the near line + 5 of the far one) x its pixels
    NfCross c;
    if (e < 2 * nRow) {                                            // rows above / below the tile, corners included
        const bool below = e >= nRow;
        const int r = below ? e - nRow : e;
        const int combo = r / kNfLW, xo = r - combo * kNfLW;
        const bool far = combo >= 10;
        const int c5 = far ? combo - 10 : combo;
        c.jx = X0 - kNlR + xo;
        c.l = c5 % 5 - 2;
        if (!below) { c.jy = Y0 - (far ? 2 : 1); c.k = far ? 2 : 1 + c5 / 5; }
        else { c.jy = Y0 + TY + (far ? 1 : 0); c.k = far ? -2 : -2 + c5 / 5; }
    } else {                                                       // columns left / right of the tile, its own rows only
        const int r0 = e - 2 * nRow;
        const bool right = r0 >= nCol;
        const int r = right ? r0 - nCol : r0;
        const int combo = r / TY, yo = r - combo * TY;
        const bool far = combo >= 10;
        const int c2 = far ? combo - 10 : combo;
        c.jy = Y0 + yo;
        c.k = far ? c2 - 2 : c2 / 2 - 2;
        const int la = far ? 2 : 1 + (c2 & 1);
        if (!right) { c.jx = X0 - (far ? 2 : 1); c.l = la; }
        else { c.jx = X0 + kNfW + (far ? 1 : 0); c.l = -la; }
    }
    const int cc = (c.k + 2) * 5 + (c.l + 2);
    c.s = cc - (cc > 12 ? 1 : 0);
    const int tx = c.jx + c.l, ty = c.jy + c.k;
    c.need = e < 2 * nRow + 2 * nCol && c.jx >= 0 && c.jx < nx && c.jy >= 0 && c.jy < ny && tx >= X0 && tx < X0 + kNfW && ty >= Y0 &&
             ty < Y0 + TY && tx < nx && ty < ny;
    if (!c.need) { c.s = 0; c.k = c.l = 0; c.jx = X0; c.jy = Y0; }   // (an idle lane still forms addresses: keep them inside the planes)
    return c;
}

// Crossing edges of a thread (at most `rounds` of them): requested before the thread's own slots are computed, evaluated for BOTH
// flow components in the first dual phase (they share the decoding, the weight and wt); the second component's results wait in
// registers for the second phase's LDS writes.
template <int TY> struct NfCrossRegs {
    static constexpr int nItems = 30 * kNfLW + 30 * TY, NT = kNfW * TY, rounds = (nItems + NT - 1) / NT;
    NfCross c[rounds];
    float p[rounds], q[rounds], w[rounds], wt[rounds];   // old duals of both components, weight, wt of the ring pixel
    int at[rounds];                                       // float index of the ring entry inside L.pq, -1 = no edge
};

template <int TY>
__device__ __forceinline__ void nf_cross_loads(const NfArgs &a, NfRsrc rs, unsigned pb, int p_old, int q_old, int X0, int Y0, NfCrossRegs<TY> &cr)
{
#pragma unroll
    for (int r = 0; r < NfCrossRegs<TY>::rounds; ++r) {
        cr.c[r] = nf_cross_item<TY>((int)threadIdx.x + r * NfCrossRegs<TY>::NT, X0, Y0, a.nx, a.ny);
        const unsigned j4 = 4u * (unsigned)(cr.c[r].jy * a.nx + cr.c[r].jx);   // (an idle lane: the tile's first pixel, slot 0)
        cr.p[r] = nf_ld(rs, 0u, (unsigned)(p_old + cr.c[r].s) * pb + j4);      // the slot differs per lane: its plane goes into the lane offset
        cr.q[r] = nf_ld(rs, 0u, (unsigned)(q_old + cr.c[r].s) * pb + j4);
        cr.w[r] = nf_ld(rs, 0u, (unsigned)(N_W + cr.c[r].s) * pb + j4);
        cr.wt[r] = nf_ld(rs, (unsigned)N_WT * pb, j4);
    }
}

// First phase: new p of the edge into the LDS ring, new q into cr.q (kept for nf_cross_store_q)
template <int TY>
__device__ __forceinline__ void nf_cross_compute(NfLds<TY> &L, const NfArgs &a, int X0, int Y0, NfCrossRegs<TY> &cr)
{
#pragma unroll
    for (int r = 0; r < NfCrossRegs<TY>::rounds; ++r) {
        const NfCross &c = cr.c[r];
        cr.at[r] = -1;
        if (c.need) {
            const int ry = c.jy - Y0 + kNlR, rx = c.jx - X0 + kNlR;
            const double rwt = recip_f64(cr.wt[r]);
            const float pn = nl_dual_update(cr.p[r], cr.w[r], L.ub[0][ry][rx], L.ub[0][ry + c.k][rx + c.l], cr.wt[r], rwt, a.tau);
            cr.q[r] = nl_dual_update(cr.q[r], cr.w[r], L.ub[1][ry][rx], L.ub[1][ry + c.k][rx + c.l], cr.wt[r], rwt, a.tau);
            cr.at[r] = (c.s * (TY + 2 * kNlR) + ry) * kNfLW + rx;
            (&L.pq[0][0][0])[cr.at[r]] = pn;
        }
    }
}
template <int TY>
__device__ __forceinline__ void nf_cross_store_q(NfLds<TY> &L, const NfCrossRegs<TY> &cr)
{
#pragma unroll
    for (int r = 0; r < NfCrossRegs<TY>::rounds; ++r)
        if (cr.at[r] >= 0) (&L.pq[0][0][0])[cr.at[r]] = cr.q[r];
}

// The dual step of one flow component for the thread's own pixel: 24 slots in batches of kNlBatch (loads of a batch requested
// together); new dual to HBM, to LDS and kept in preg.  The first component also loads the 24 weights (kept in wreg).
template <int TY, bool BORDER, int COMP>
__device__ __forceinline__ void nf_dual_own(NfLds<TY> &L, const NfArgs &a, NfRsrc rs, unsigned pb, int d_old, int d_new, int X0, int Y0, int lx, int ly,
                                            unsigned i4, float wt, double rwt, float (&wreg)[kNlSlots], float (&preg)[kNlSlots],
                                            bool (&oks)[kNlSlots])
{
    // pb = bytes per plane; d_old / d_new = first plane of the component's 24 duals in the buffer read / written
    const int x = X0 + lx, y = Y0 + ly;
    const float ai = L.ub[COMP][ly + kNlR][lx + kNlR];
#pragma unroll
    for (int s0 = 0; s0 < kNlSlots; s0 += kNlBatch) {
#pragma unroll
        for (int s = s0; s < s0 + kNlBatch; ++s) {
            preg[s] = nf_ld(rs, (unsigned)(d_old + s) * pb, i4);
            if (COMP == 0) {
                const int k = nl_slot_k(s), l = nl_slot_l(s);
                wreg[s] = nf_ld(rs, (unsigned)(N_W + s) * pb, i4);
                oks[s] = !BORDER || (x + l >= 0 && x + l < a.nx && y + k >= 0 && y + k < a.ny);
            }
        }
        // (a slot that leaves the image is evaluated like the others -- its weight and its neighbour's ubar read as 0 -- but never
        // stored: such slots keep their duals, :1145, and the divergence skips them)
        float pb6[kNlBatch], wb6[kNlBatch], nb6[kNlBatch];
#pragma unroll
        for (int q = 0; q < kNlBatch; ++q) {
            const int s = s0 + q;
            pb6[q] = preg[s];
            wb6[q] = wreg[s];
            nb6[q] = L.ub[COMP][ly + kNlR + nl_slot_k(s)][lx + kNlR + nl_slot_l(s)];
        }
        nl_dual_update_n<kNlBatch>(pb6, wb6, ai, nb6, wt, rwt, a.tau);
#pragma unroll
        for (int q = 0; q < kNlBatch; ++q) {
            const int s = s0 + q;
            if (!(kNfWhatIf & 16) && (!BORDER || oks[s])) nf_st(rs, (unsigned)(d_new + s) * pb, i4, pb6[q]);
            L.pq[s][ly + kNlR][lx + kNlR] = pb6[q];
            preg[s] = pb6[q];
        }
    }
}

// non_local_divergence (:1056-1079) of one component from registers (own duals, weights) and LDS (mirror duals), slot order 0..23.
template <int TY, bool BORDER>
__device__ __forceinline__ float nf_divergence(const NfLds<TY> &L, int lx, int ly, const float (&wreg)[kNlSlots],
                                               const float (&preg)[kNlSlots], const bool (&oks)[kNlSlots])
{
    float d = 0.0f;
#pragma unroll
    for (int s = 0; s < kNlSlots; ++s) {
        const int k = nl_slot_k(s), l = nl_slot_l(s);
        const float m = L.pq[kNlSlots - 1 - s][ly + kNlR + k][lx + kNlR + l];
        const float nd = d + wreg[s] * (preg[s] - m);
        d = (!BORDER || oks[s]) ? nd : d;
    }
    return d;
}

// csad_select of nltv_kernels.hpp with the 7x7 neighbourhoods of I0 and I1w read from the staged LDS tiles.
template <int TY, bool BORDER>
__device__ __forceinline__ float nf_csad_select(const NfLds<TY> &L, int lx, int ly, int x, int y, int nx, int ny, float t1, float t2,
                                                float r, float sg, double rsg, float l_t, const float (&kl)[kCsSlots])
{
    const float i0 = L.im[0][ly + kCsR][lx + kCsR], i1w = L.im[1][ly + kCsR][lx + kCsR];
    float A[kCsSlots];
    int n = 0;
#pragma unroll
    for (int s = 0; s < kCsSlots; ++s) {
        const int c = s < 24 ? s : s + 1;
        const int k = c / 7 - kCsR, l = c % 7 - kCsR;
        const bool ok = !BORDER || (x + l >= 0 && x + l < nx && y + k >= 0 && y + k < ny);
        const float a0 = L.im[0][ly + kCsR + k][lx + kCsR + l], a1 = L.im[1][ly + kCsR + k][lx + kCsR + l];
        // (an out-of-image slot reads the staged zeros; its term is discarded below, as the two-kernel path discards its own)
        const float bs = (kNfWhatIf & 32) ? (i0 - a0 - i1w + a1 + t1 + t2) * sg : div_by(i0 - a0 - i1w + a1 + t1 + t2, sg, rsg);   // :1716-1717
        A[s] = ok ? -(bs - r) : __builtin_inff();
        n += ok ? 1 : 0;
    }
    if (!(kNfWhatIf & 8)) sort48(A);
    float m = __builtin_inff();
#pragma unroll
    for (int c = 1; c <= kCsSlots; ++c) {
        const float bv = !BORDER ? kl[c - 1] * sg : (c <= n + 1) ? (float)(n + 2 - 2 * c) * l_t * sg : __builtin_inff();
        m = __builtin_fminf(m, __builtin_fmaxf(A[c - 1], bv));
    }
    return m;
}

template <bool DATA_L1, int TY, bool BORDER>
__device__ __forceinline__ void nl_fused_body(const NfArgs &a, NfLds<TY> &L, int X0, int Y0)
{
    constexpr int NT = kNfW * TY;
    const int nx = a.nx, ny = a.ny;
    const int t = threadIdx.x, lx = t & 63, ly = t >> 6;
    const int x = X0 + lx, y = Y0 + ly;
    const bool in = x < nx && y < ny;
    const unsigned i4 = in ? 4u * (unsigned)(y * nx + x) : 0u;
    const unsigned pb = (unsigned)a.pe * 4u;
    const NfRsrc rs = __builtin_amdgcn_make_buffer_rsrc(a.base, 0, (unsigned)N_COUNT * pb, 0x00020000);
    const int cur = a.cur;
    const int ub_old = cur ? N_UBB1 : N_UB1, ub_new = cur ? N_UB1 : N_UBB1;          // two consecutive planes each
    const int p_old = cur ? N_PB : N_P, p_new = cur ? N_P : N_PB, q_old = cur ? N_QB : N_Q, q_new = cur ? N_Q : N_QB;

    // ---- stage ubar (ring 2) and I0 / I1w (ring 3); outside the image: 0 (never used in an accepted term)
    {
        constexpr int RW = kNfLW, RH = TY + 2 * kNlR, n_ub = 2 * RH * RW;
#pragma unroll
        for (int q = 0; q < (n_ub + NT - 1) / NT; ++q) {
            const int e = t + q * NT;
            const int comp = e / (RH * RW), rem = e - comp * (RH * RW), ry = rem / RW, rx = rem - ry * RW;
            const int gx = X0 - kNlR + rx, gy = Y0 - kNlR + ry;
            const bool ok = e < n_ub && gx >= 0 && gx < nx && gy >= 0 && gy < ny;
            const float v = nf_ld(rs, 0u, ok ? (unsigned)(ub_old + comp) * pb + 4u * (unsigned)(gy * nx + gx) : 0u);
            if (e < n_ub) L.ub[comp][ry][rx] = ok ? v : 0.0f;
        }
        if (!DATA_L1) {
            constexpr int IW = kNfIW, IH = TY + 2 * kCsR, n_im = 2 * IH * IW;
#pragma unroll
            for (int q = 0; q < (n_im + NT - 1) / NT; ++q) {
                const int e = t + q * NT;
                const int comp = e / (IH * IW), rem = e - comp * (IH * IW), ry = rem / IW, rx = rem - ry * IW;
                const int gx = X0 - kCsR + rx, gy = Y0 - kCsR + ry;
                const bool ok = e < n_im && gx >= 0 && gx < nx && gy >= 0 && gy < ny;
                const float v = nf_ld(rs, 0u, ok ? (unsigned)(comp ? N_I1W : N_I0) * pb + 4u * (unsigned)(gy * nx + gx) : 0u);
                if (e < n_im) L.im[comp][ry][rx] = ok ? v : 0.0f;
            }
        }
    }
    const float wt = nf_ld(rs, (unsigned)N_WT * pb, i4);
    const double rwt = recip_f64(wt);            // wt > 0 (every pixel has valid neighbours): shared by 48 quotients
    __syncthreads();

    // ---- data step (nl_primal_kernel of nltv_kernels.hpp, first half): v from u and the warp's constants -- arithmetic on LDS and
    // six values of the pixel, independent of the duals.  It runs first: its 48 sorted terms are dead before the 48 weight / dual
    // registers come alive.  (Measured and dropped: running it last on every second tile, chessboard-wise, so that half of the CUs
    // compute while the other half stream the dual planes -- no faster; requesting a phase's 48 loads at once -- slower.)
    float u1k = 0.0f, u2k = 0.0f, v1 = 0.0f, v2 = 0.0f;
    if (in) {
        u1k = nf_ld(rs, (unsigned)N_U1 * pb, i4);
        u2k = nf_ld(rs, (unsigned)N_U2 * pb, i4);
        const float grad = nf_ld(rs, (unsigned)N_GRAD * pb, i4);
        v1 = u1k;                                                        // :1735-1736
        v2 = u2k;
        if (DATA_L1) {
            const float ixw = nf_ld(rs, (unsigned)N_I1WX * pb, i4), iyw = nf_ld(rs, (unsigned)N_I1WY * pb, i4);
            threshold_step(nf_ld(rs, (unsigned)N_I1W * pb, i4), ixw, iyw, grad, recip_f64(grad), u1k, u2k, a.l_t, v1, v2);
        } else if (!(kNfWhatIf & 1) && !(grad < grad_zero_bound())) {
            const float ixw = nf_ld(rs, (unsigned)N_I1WX * pb, i4), iyw = nf_ld(rs, (unsigned)N_I1WY * pb, i4);
            // quotients by sg (>= 1e-4), wt and theta below: div_by returns the IEEE quotient's bits for a finite numerator and a
            // finite divisor > 0 (faldoi_device.hpp), with the reciprocal shared
            const float sg = __builtin_sqrtf(grad);
            const double rsg = recip_f64(sg);
            const float r = div_by(ixw * u1k + iyw * u2k, sg, rsg);
            const float m = nf_csad_select<TY, BORDER>(L, lx, ly, x, y, nx, ny, nf_ld(rs, (unsigned)N_T1 * pb, i4),
                                                       nf_ld(rs, (unsigned)N_T2 * pb, i4), r, sg, rsg, a.l_t, a.kl);
            v1 = u1k - div_by(ixw * m, sg, rsg);                         // :1756-1757
            v2 = u2k - div_by(iyw * m, sg, rsg);
        }
        asm volatile("" : "+v"(v1), "+v"(v2));                           // (keeps the scheduler from moving the step into the dual phases)
    }
    __builtin_amdgcn_sched_barrier(0);

    float wreg[kNlSlots], preg[kNlSlots];
    bool oks[kNlSlots];
    NfCrossRegs<TY> cr;
    if (!(kNfWhatIf & 2)) nf_cross_loads<TY>(a, rs, pb, p_old, q_old, X0, Y0, cr);       // in flight while the own slots are computed
    if (in) nf_dual_own<TY, BORDER, 0>(L, a, rs, pb, p_old, p_new, X0, Y0, lx, ly, i4, wt, rwt, wreg, preg, oks);
    if (!(kNfWhatIf & 2)) nf_cross_compute<TY>(L, a, X0, Y0, cr);
    __syncthreads();
    float dp = in ? nf_divergence<TY, BORDER>(L, lx, ly, wreg, preg, oks) : 0.0f;
    __syncthreads();
    if (in) nf_dual_own<TY, BORDER, 1>(L, a, rs, pb, q_old, q_new, X0, Y0, lx, ly, i4, wt, rwt, wreg, preg, oks);
    if (!(kNfWhatIf & 2)) nf_cross_store_q<TY>(L, cr);
    __syncthreads();
    if (!in) return;
    float dq = nf_divergence<TY, BORDER>(L, lx, ly, wreg, preg, oks);

    // ---- ofnltv_getP (:1090-1120) + relaxation
    dp = div_by(dp, wt, rwt);
    dq = div_by(dq, wt, rwt);
    const float n1 = u1k - a.tau * (dp + div_by(u1k - v1, a.theta, a.rtheta));   // :1106-1107, note +div
    const float n2 = u2k - a.tau * (dq + div_by(u2k - v2, a.theta, a.rtheta));
    nf_st(rs, (unsigned)N_U1 * pb, i4, n1);
    nf_st(rs, (unsigned)N_U2 * pb, i4, n2);
    nf_st(rs, (unsigned)ub_new * pb, i4, 2.0f * n1 - u1k);               // :1774-1777
    nf_st(rs, (unsigned)(ub_new + 1) * pb, i4, 2.0f * n2 - u2k);
}

template <bool DATA_L1, int TY>
__global__ __launch_bounds__(kNfW * TY) void nl_fused_kernel(NfArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char nf_lds_raw[];
    NfLds<TY> &L = *reinterpret_cast<NfLds<TY> *>(nf_lds_raw);
    // XCD-aware tile order: blocks are dealt to the 8 XCDs round-robin, so XCD k gets the `per` consecutive tiles [k per, (k+1) per)
    // (neighbouring tiles re-read each other's ring from one L2), block j of the XCD takes the j-th of them.  Among those the tiles on
    // the image frame go FIRST: they are the slower ones (stencil validity selects), and the grid runs in rounds of one tile per CU --
    // a slow tile in the last round is what the whole launch waits for.
    const int per = gridDim.x >> 3, ntiles = a.ntx * a.nty;
    const int k = blockIdx.x & 7, j = blockIdx.x >> 3;
    int tile = k * per + j;
    if (per <= 64) {
        const int lane = threadIdx.x & 63;
        const int tq = k * per + lane;                                 // lane q looks at the q-th tile of this XCD
        const bool valid = lane < per && tq < ntiles;
        const int qy = tq / a.ntx, qx = tq - qy * a.ntx;
        const bool inner = qx * kNfW >= kCsR && qy * TY >= kCsR && qx * kNfW + kNfW + kCsR <= a.nx && qy * TY + TY + kCsR <= a.ny;
        const unsigned long long frame = __builtin_amdgcn_ballot_w64(valid && !inner), rest = __builtin_amdgcn_ballot_w64(valid && inner);
        const int nframe = __builtin_popcountll(frame);
        const unsigned long long from = j < nframe ? frame : rest;
        const int want = j < nframe ? j : j - nframe;                  // the want-th set bit of `from`
        const int rank = __builtin_popcountll(from & ((1ull << lane) - 1ull));
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(((from >> lane) & 1ull) != 0ull && rank == want);
        tile = hit ? k * per + __builtin_ctzll(hit) : ntiles;          // (no such tile: this block has nothing to do)
    }
    if (tile >= ntiles) return;
    const int tyi = tile / a.ntx, txi = tile - tyi * a.ntx;
    const int X0 = txi * kNfW, Y0 = tyi * TY;
    const bool interior = X0 >= kCsR && Y0 >= kCsR && X0 + kNfW + kCsR <= a.nx && Y0 + TY + kCsR <= a.ny;
    if (interior) nl_fused_body<DATA_L1, TY, false>(a, L, X0, Y0);
    else nl_fused_body<DATA_L1, TY, true>(a, L, X0, Y0);
}

}  // namespace faldoi
