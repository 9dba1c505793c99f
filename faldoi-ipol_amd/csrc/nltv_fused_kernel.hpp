// nltv_fused_kernel.hpp -- one launch per inner iteration of the NLTV solvers (NLTV-CSAD, BASELINE configs[4]; NLTV-L1).
//
// The two streaming kernels of nltv_kernels.hpp write the 48 dual planes and read them straight back (the divergence needs, for
// every slot s of a pixel, the MIRROR dual p[23-s] of the neighbour the slot points at).  Here a workgroup owns a tile of
// 64 x TY pixels and keeps the new duals of the tile in LDS between the two steps:
//
//   stage      ubar (2-pixel ring) and, for the CSAD data term, I0 / I1w (3-pixel ring) into LDS
//   data       CSAD selection (:1733-1758; six probes of the warp's rank table, neighbours from LDS) or TV-L1 thresholding: v from u
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
// LDS: 24 x (TY+4) x 68 floats of duals + the staged planes = 154.5 KB at TY = 16 (of 160): one workgroup of 64 x TY threads per CU.
#pragma once
#include "nltv_kernels.hpp"

#pragma clang fp contract(off)

// FALDOI_NFWHATIF (timing-only builds, wrong results; tools/nl_whatif.sh): 1 no data step | 2 no crossing edges | 4 trivial dual
// arithmetic | 16 no HBM stores of the duals | 64 the weights of slots 12..23 not loaded | 128 rank words 9..11 not loaded
#ifndef FALDOI_NFWHATIF
#define FALDOI_NFWHATIF 0
#endif


namespace faldoi {

constexpr int kNfWhatIf = FALDOI_NFWHATIF;

// FALDOI_NL_CHECK=1 (the second library build.py makes, libfaldoi_hip_check.so; tests/test_gpu_round4.py runs the geometry sweep on it
// once): every LDS index and every slab offset this kernel forms is compared with the extent it must stay inside -- the array of the
// LDS struct, the pixel range of a plane, the slab -- and a violation raises a bit of g_nf_check[0] (which check) and records the
// offending value; the access itself is then made at index 0.  In the product build the checks compile to nothing.
#ifndef FALDOI_NL_CHECK
#define FALDOI_NL_CHECK 0
#endif
constexpr bool kNfCheck = FALDOI_NL_CHECK != 0;
enum NfCheckBit { NFC_PQ = 0, NFC_UB, NFC_IM, NFC_PIXEL, NFC_SLAB, NFC_PLANE, NFC_STAGE_LDS, NFC_CROSS_AT, NFC_RANK_BYTE };
__device__ unsigned g_nf_check[4];            // [0] bits of NfCheckBit, [1] number of violations, [2] largest offending value, [3] launches checked
__device__ __forceinline__ bool nf_check(bool ok, int bit, unsigned value)
{
    if (kNfCheck && !ok) {
        atomicOr(&g_nf_check[0], 1u << bit);
        atomicAdd(&g_nf_check[1], 1u);
        atomicMax(&g_nf_check[2], value);
    }
    return !kNfCheck || ok;
}
constexpr int kNfW = 64;                      // tile width: one wavefront per tile row
constexpr int kNfLW = kNfW + 2 * kNlR;        // 68: LDS row of a dual plane / of ubar (2-pixel ring)
constexpr int kNfIW = kNfW + 8;                // 72: LDS row of the staged I0 / I1w (3-pixel ring, 70 used): a window position byte
                                              // e = (k+3)*8 + (l+3) of the rank table addresses the row-major tile as 9 e - 8 (e & 7)

template <int TY> struct NfLds {
    float pq[kNlSlots][TY + 2 * kNlR][kNfLW];
    float ub[2][TY + 2 * kNlR][kNfLW];
    float im[2][TY + 2 * kCsR][kNfIW];
    // every access goes through these (FALDOI_NL_CHECK: range-checked, see nf_check; otherwise plain indexing)
    __device__ __forceinline__ float &pq_at(int s, int r, int c);
    __device__ __forceinline__ float &pq_flat(int at);
    __device__ __forceinline__ float &ub_at(int comp, int r, int c);
    __device__ __forceinline__ float &im_at(int comp, int r, int c);
};

template <int TY> __device__ __forceinline__ float &NfLds<TY>::pq_at(int s, int r, int c)
{
    const bool ok = nf_check(s >= 0 && s < kNlSlots && r >= 0 && r < TY + 2 * kNlR && c >= 0 && c < kNfLW, NFC_PQ, (unsigned)((s * 64 + r) * 128 + c));
    return ok ? pq[s][r][c] : pq[0][0][0];
}
template <int TY> __device__ __forceinline__ float &NfLds<TY>::pq_flat(int at)
{
    const bool ok = nf_check(at >= 0 && at < kNlSlots * (TY + 2 * kNlR) * kNfLW, NFC_CROSS_AT, (unsigned)at);
    return (&pq[0][0][0])[ok ? at : 0];
}
template <int TY> __device__ __forceinline__ float &NfLds<TY>::ub_at(int comp, int r, int c)
{
    const bool ok = nf_check(comp >= 0 && comp < 2 && r >= 0 && r < TY + 2 * kNlR && c >= 0 && c < kNfLW, NFC_UB, (unsigned)((comp * 64 + r) * 128 + c));
    return ok ? ub[comp][r][c] : ub[0][0][0];
}
template <int TY> __device__ __forceinline__ float &NfLds<TY>::im_at(int comp, int r, int c)
{
    const bool ok = nf_check(comp >= 0 && comp < 2 && r >= 0 && r < TY + 2 * kCsR && c >= 0 && c < kNfW + 2 * kCsR, NFC_IM, (unsigned)((comp * 64 + r) * 128 + c));
    return ok ? im[comp][r][c] : im[0][0][0];
}

struct NfArgs {
    float *base;
    size_t pe;
    int nx, ny, ntx, nty;
    int cur;                                  // which buffer of the duals / ubar holds the state this iteration reads
    float tau, l_t, theta;
    double rtheta;                            // 1 / theta in double (host), for the exact quotients by theta
    int check_selftest;                       // check build, FALDOI_NL_CHECK_SELFTEST=1: one deliberately out-of-range index per kind (the
                                              // detector must not be vacuous)
};

// Plane accesses go through ONE buffer resource over the whole slab: the per-lane byte offset of a pixel sits in one VGPR and the
// plane's offset (uniform) in an SGPR, instead of a 64-bit address per access (host: the slab is below 4 GB, else the two-kernel
// path runs).
typedef __amdgpu_buffer_rsrc_t NfRsrc;
// (check build: g_nf_extent = { bytes of a plane's pixels, bytes of a plane, bytes of the slab }, set by the kernel's first statement)
__device__ unsigned g_nf_extent[3];
__device__ __forceinline__ unsigned nf_checked_off(unsigned plane_bytes, unsigned voff)
{
    if (!kNfCheck) return voff;
    bool ok = nf_check((voff & 3u) == 0u && (plane_bytes % g_nf_extent[1]) == 0u, NFC_PLANE, plane_bytes);
    // a pixel of a plane: the plane sits in the scalar offset, or (crossing edges) is folded into the lane offset
    ok &= nf_check((plane_bytes != 0u ? voff : voff % g_nf_extent[1]) + 4u <= g_nf_extent[0], NFC_PIXEL, voff);
    ok &= nf_check((unsigned long long)plane_bytes + voff + 4ull <= g_nf_extent[2], NFC_SLAB, voff);
    return ok ? voff : 0u;
}
__device__ __forceinline__ float nf_ld(NfRsrc r, unsigned plane_bytes, unsigned voff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, nf_checked_off(plane_bytes, voff), plane_bytes, 0));
}
__device__ __forceinline__ unsigned nf_ld_u32(NfRsrc r, unsigned plane_bytes, unsigned voff)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, nf_checked_off(plane_bytes, voff), plane_bytes, 0);
}
__device__ __forceinline__ void nf_st(NfRsrc r, unsigned plane_bytes, unsigned voff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, nf_checked_off(plane_bytes, voff), plane_bytes, 0);
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
            const float pn = nl_dual_update(cr.p[r], cr.w[r], L.ub_at(0, ry, rx), L.ub_at(0, ry + c.k, rx + c.l), cr.wt[r], rwt, a.tau);
            cr.q[r] = nl_dual_update(cr.q[r], cr.w[r], L.ub_at(1, ry, rx), L.ub_at(1, ry + c.k, rx + c.l), cr.wt[r], rwt, a.tau);
            cr.at[r] = (c.s * (TY + 2 * kNlR) + ry) * kNfLW + rx;
            L.pq_flat(cr.at[r]) = pn;
        }
    }
}
template <int TY>
__device__ __forceinline__ void nf_cross_store_q(NfLds<TY> &L, const NfCrossRegs<TY> &cr)
{
#pragma unroll
    for (int r = 0; r < NfCrossRegs<TY>::rounds; ++r)
        if (cr.at[r] >= 0) L.pq_flat(cr.at[r]) = cr.q[r];
}

// The dual step of one flow component for the thread's own pixel: 24 slots in batches of kNlBatch.  A batch's loads are REQUESTED
// one batch ahead of its arithmetic (nf_batch_loads / nf_batch_update, interleaved by nl_fused_body): a workgroup's sixteen waves
// run through the phases in step (barriers), so a load that is waited for right after its request costs its whole latency on every
// wave at once -- round 3's kernel spent about half of a tile's time that way.  New dual to HBM, to LDS and kept in preg; the first
// component also loads the 24 weights (kept in wreg).
struct NfBatch {
    float p[kNlBatch], w[kNlBatch];
};
template <int COMP>
__device__ __forceinline__ void nf_batch_loads(NfRsrc rs, unsigned pb, int d_old, int s0, unsigned i4, NfBatch &b)
{
    // pb = bytes per plane; d_old = first plane of the component's 24 duals in the buffer that is read
#pragma unroll
    for (int q = 0; q < kNlBatch; ++q) {
        b.p[q] = nf_ld(rs, (unsigned)(d_old + s0 + q) * pb, i4);
        if (COMP == 0) b.w[q] = ((kNfWhatIf & 64) && s0 + q >= 12) ? 0.5f : nf_ld(rs, (unsigned)(N_W + s0 + q) * pb, i4);
    }
}
template <int TY, bool BORDER, int COMP>
__device__ __forceinline__ void nf_batch_update(NfLds<TY> &L, const NfArgs &a, NfRsrc rs, unsigned pb, int d_new, int s0, int x, int y, int lx, int ly,
                                                unsigned i4, float wt, double rwt, const NfBatch &b, float (&wreg)[kNlSlots],
                                                float (&preg)[kNlSlots], bool (&oks)[kNlSlots])
{
    const float ai = L.ub_at(COMP, ly + kNlR, lx + kNlR);
    // (a slot that leaves the image is evaluated like the others -- its weight and its neighbour's ubar read as 0 -- but never
    // stored: such slots keep their duals, :1145, and the divergence skips them)
    float pb6[kNlBatch], wb6[kNlBatch], nb6[kNlBatch];
#pragma unroll
    for (int q = 0; q < kNlBatch; ++q) {
        const int s = s0 + q;
        if (COMP == 0) {
            const int k = nl_slot_k(s), l = nl_slot_l(s);
            wreg[s] = b.w[q];
            oks[s] = !BORDER || (x + l >= 0 && x + l < a.nx && y + k >= 0 && y + k < a.ny);
        }
        pb6[q] = b.p[q];
        wb6[q] = wreg[s];
        nb6[q] = L.ub_at(COMP, ly + kNlR + nl_slot_k(s), lx + kNlR + nl_slot_l(s));
    }
    nl_dual_update_n<kNlBatch>(pb6, wb6, ai, nb6, wt, rwt, a.tau);
#pragma unroll
    for (int q = 0; q < kNlBatch; ++q) {
        const int s = s0 + q;
        if (!(kNfWhatIf & 16) && (!BORDER || oks[s])) nf_st(rs, (unsigned)(d_new + s) * pb, i4, pb6[q]);
        L.pq_at(s, ly + kNlR, lx + kNlR) = pb6[q];
        preg[s] = pb6[q];
    }
}
// Batches 1..3 of a component (batch 0 was requested by the caller, earlier): request the next batch, then update the current one.
template <int TY, bool BORDER, int COMP>
__device__ __forceinline__ void nf_dual_own(NfLds<TY> &L, const NfArgs &a, NfRsrc rs, unsigned pb, int d_old, int d_new, int x, int y, int lx, int ly,
                                            unsigned i4, bool in, float wt, double rwt, NfBatch &b0, float (&wreg)[kNlSlots],
                                            float (&preg)[kNlSlots], bool (&oks)[kNlSlots])
{
    NfBatch nb[kNlSlots / kNlBatch];
    nb[0] = b0;
#pragma unroll
    for (int k = 0; k < kNlSlots / kNlBatch; ++k) {
        if (k + 1 < kNlSlots / kNlBatch) {
            nf_batch_loads<COMP>(rs, pb, d_old, (k + 1) * kNlBatch, i4, nb[k + 1]);
            __builtin_amdgcn_sched_barrier(0);                           // (the requests stay ahead of the arithmetic below)
        }
        if (in) nf_batch_update<TY, BORDER, COMP>(L, a, rs, pb, d_new, k * kNlBatch, x, y, lx, ly, i4, wt, rwt, nb[k], wreg, preg, oks);
    }
}

// non_local_divergence (:1056-1079) of one component from registers (own duals, weights) and LDS (mirror duals), slot order 0..23.
template <int TY, bool BORDER>
__device__ __forceinline__ float nf_divergence(NfLds<TY> &L, int lx, int ly, const float (&wreg)[kNlSlots],
                                               const float (&preg)[kNlSlots], const bool (&oks)[kNlSlots])
{
    float d = 0.0f;
#pragma unroll
    for (int s = 0; s < kNlSlots; ++s) {
        const int k = nl_slot_k(s), l = nl_slot_l(s);
        const float m = L.pq_at(kNlSlots - 1 - s, ly + kNlR + k, lx + kNlR + l);
        const float nd = d + wreg[s] * (preg[s] - m);
        d = (!BORDER || oks[s]) ? nd : d;
    }
    return d;
}

// csad_select_ranked of nltv_kernels.hpp with the neighbours of I0 and I1w read from the staged LDS tiles: six probes per pixel.
template <int TY> struct NfCsadFetch {
    const float *im;          // &L.im[0][ly][lx]: window position (k, l) = (-3, -3) of this pixel
    int lx, ly;               // (check build only)
    __device__ __forceinline__ void operator()(unsigned e, bool, float &a0, float &a1) const
    {
        // (a probe that is not live carries a byte of the table all the same: 0 <= e <= 54, inside the staged tile)
        bool ok = nf_check((e >> 3) <= 2u * kCsR && (e & 7u) <= 2u * kCsR, NFC_RANK_BYTE, e);
        ok = ok && nf_check(ly + (int)(e >> 3) < TY + 2 * kCsR && lx + (int)(e & 7u) < kNfW + 2 * kCsR, NFC_IM, e);
        const unsigned off = ok ? 9u * e - 8u * (e & 7u) : 0u;
        a0 = im[off];
        a1 = im[off + (TY + 2 * kCsR) * kNfIW];
    }
};
template <int TY, bool BORDER>
__device__ __forceinline__ float nf_csad_select(NfLds<TY> &L, const unsigned (&rk)[kCsRankWords], int lx, int ly, int x, int y, int nx,
                                                int ny, float t1, float t2, float r, float sg, double rsg, float l_t)
{
    const float i0 = L.im_at(0, ly + kCsR, lx + kCsR), i1w = L.im_at(1, ly + kCsR, lx + kCsR);
    const int n = BORDER ? csad_valid_count(x, y, nx, ny) : kCsSlots;
    return csad_select_ranked<!BORDER>(rk, n, NfCsadFetch<TY>{ &L.im[0][ly][lx], lx, ly }, i0, i1w, t1, t2, r, sg, rsg, l_t);
}

typedef float nf_f4 __attribute__((ext_vector_type(4)));
typedef unsigned nf_u4 __attribute__((ext_vector_type(4)));

// One staged quad: four consecutive pixels of a row of the tile's footprint, one 16-byte load (4-byte aligned: gfx950 takes it),
// one 16-byte LDS write (rows of 68 / 72 floats keep the quads aligned).  Pixels outside the image become 0.
struct NfQuad {
    nf_f4 v;
    int lds;              // float index of the quad inside its LDS array, -1: this thread stages nothing
};
template <int ROWS, int ROW_FLOATS, int RING>
__device__ __forceinline__ NfQuad nf_stage_load(NfRsrc rs, unsigned pb, int plane0, int t, int X0, int Y0, int nx, int ny)
{
    constexpr int Q = ROW_FLOATS / 4, n = 2 * ROWS * Q;
    const int comp = t / (ROWS * Q), rem = t - comp * (ROWS * Q), ry = rem / Q, rq = rem - ry * Q;
    const int gx = X0 - RING + 4 * rq, gy = Y0 - RING + ry;
    const bool row = t < n && gy >= 0 && gy < ny && gx < nx;
    // The tile's first quad of a row starts RING pixels left of the image when the tile does (X0 = 0): it is loaded from column 0 and
    // shifted -- on row 0 of the slab's first plane the unshifted address would lie before the buffer, and the whole load returns 0.
    // A quad that straddles the image's right edge reads the next row's start (or, at the slab's very end, outside the buffer
    // resource, which returns 0); those components are zeroed below.
    const bool lead = gx < 0;                                          // then gx == -RING
    const unsigned off = row ? (unsigned)(plane0 + comp) * pb + 4u * (unsigned)(gy * nx + (lead ? 0 : gx)) : 0u;
    // (check build: a quad may run past its row -- masked below -- but never past the slab)
    const bool inside = nf_check(!kNfCheck || (unsigned long long)off + 16ull <= g_nf_extent[2], NFC_SLAB, off);
    const nf_u4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, inside ? off : 0u, 0u, 0);
    NfQuad q;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // (elements are copied to scalars first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 for
        // every index with this compiler -- found in the ISA, where the 16-byte load had shrunk to one dword)
        const unsigned e = raw[j], es = raw[j >= RING ? j - RING : 0];
        const bool ok = row && (lead ? j >= RING && j - RING < nx : gx + j < nx);
        q.v[j] = ok ? __uint_as_float(lead ? es : e) : 0.0f;
    }
    q.lds = t < n ? (comp * ROWS + ry) * ROW_FLOATS + 4 * rq : -1;
    return q;
}

template <bool DATA_L1, int TY, bool BORDER>
__device__ __forceinline__ void nl_fused_body(const NfArgs &a, NfLds<TY> &L, int X0, int Y0)
{
    constexpr int NT = kNfW * TY;
    static_assert(2 * (TY + 2 * kNlR) * (kNfLW / 4) <= NT && 2 * (TY + 2 * kCsR) * (kNfIW / 4) <= NT, "one staged quad per thread");
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

    // ---- every load that does not depend on LDS is requested before anything is waited for (a thread outside the image reads
    // pixel 0 and uses nothing of it): the staged rings first (they are needed first, and loads return in order), then the pixel's
    // own values, the warp's rank table, and the first batch of duals and weights
    const NfQuad sq_ub = nf_stage_load<TY + 2 * kNlR, kNfLW, kNlR>(rs, pb, ub_old, t, X0, Y0, nx, ny);
    NfQuad sq_im;
    sq_im.lds = -1;
    if (!DATA_L1) {
        // (the helper reads plane0 + component; I0 and I1w are not neighbouring planes, so plane0 is chosen per component)
        constexpr int IH = TY + 2 * kCsR;
        const int comp = t / (IH * (kNfIW / 4));
        sq_im = nf_stage_load<IH, kNfIW, kCsR>(rs, pb, comp ? N_I1W - 1 : N_I0, t, X0, Y0, nx, ny);
    }
    const float wt = nf_ld(rs, (unsigned)N_WT * pb, i4);
    const float u1k = nf_ld(rs, (unsigned)N_U1 * pb, i4), u2k = nf_ld(rs, (unsigned)N_U2 * pb, i4);
    const float grad = nf_ld(rs, (unsigned)N_GRAD * pb, i4);
    const float ixw = nf_ld(rs, (unsigned)N_I1WX * pb, i4), iyw = nf_ld(rs, (unsigned)N_I1WY * pb, i4);
    float c0 = 0.0f, c1 = 0.0f;                    // NLTV-L1: rho_c; NLTV-CSAD: the two warp-start products inside b
    unsigned rk[kCsRankWords];
    if (DATA_L1) {
        c0 = nf_ld(rs, (unsigned)N_I1W * pb, i4);
    } else {
        c0 = nf_ld(rs, (unsigned)N_T1 * pb, i4);
        c1 = nf_ld(rs, (unsigned)N_T2 * pb, i4);
#pragma unroll
        for (int w = 0; w < kCsRankWords; ++w)
            rk[w] = ((kNfWhatIf & 128) && w >= 9) ? 0x09090909u : nf_ld_u32(rs, (unsigned)(N_RK + w) * pb, i4);
    }
    NfBatch bp0;
    nf_batch_loads<0>(rs, pb, p_old, 0, i4, bp0);
    NfCrossRegs<TY> cr;
    if (!(kNfWhatIf & 2)) nf_cross_loads<TY>(a, rs, pb, p_old, q_old, X0, Y0, cr);       // in flight during the data step and the own slots
    __builtin_amdgcn_sched_barrier(0);

    // ---- stage ubar (ring 2) and I0 / I1w (ring 3)
    if (sq_ub.lds >= 0 && nf_check(sq_ub.lds + 4 <= 2 * (TY + 2 * kNlR) * kNfLW && (sq_ub.lds & 3) == 0, NFC_STAGE_LDS, (unsigned)sq_ub.lds))
        *reinterpret_cast<nf_f4 *>(&L.ub[0][0][0] + sq_ub.lds) = sq_ub.v;
    if (!DATA_L1 && sq_im.lds >= 0 && nf_check(sq_im.lds + 4 <= 2 * (TY + 2 * kCsR) * kNfIW && (sq_im.lds & 3) == 0, NFC_STAGE_LDS, (unsigned)sq_im.lds))
        *reinterpret_cast<nf_f4 *>(&L.im[0][0][0] + sq_im.lds) = sq_im.v;
    const double rwt = recip_f64(wt);            // wt > 0 (every pixel has valid neighbours): shared by 48 quotients
    __syncthreads();

    // ---- data step (nl_primal_kernel of nltv_kernels.hpp, first half): v from u and the warp's constants -- six probes of the
    // staged tiles (CSAD) or the thresholding (L1); the first batch of the dual step is in flight meanwhile
    float v1 = u1k, v2 = u2k;                                             // :1735-1736
    if (DATA_L1) {
        threshold_step(c0, ixw, iyw, grad, recip_f64(grad), u1k, u2k, a.l_t, v1, v2);
    } else if (!(kNfWhatIf & 1) && wave_any(!(grad < grad_zero_bound()))) {
        // quotients by sg (>= 1e-4), wt and theta below: div_by returns the IEEE quotient's bits for a finite numerator and a
        // finite divisor > 0 (faldoi_device.hpp), with the reciprocal shared.  (A pixel below the bound keeps v = u; its lanes
        // evaluate the step on sg = 1 and discard it.)
        const bool live = !(grad < grad_zero_bound());
        const float sg = live ? __builtin_sqrtf(grad) : 1.0f;
        const double rsg = recip_f64(sg);
        const float r = div_by(ixw * u1k + iyw * u2k, sg, rsg);
        const float m = nf_csad_select<TY, BORDER>(L, rk, lx, ly, x, y, nx, ny, c0, c1, r, sg, rsg, a.l_t);
        const float w1 = u1k - div_by(ixw * m, sg, rsg);                  // :1756-1757
        const float w2 = u2k - div_by(iyw * m, sg, rsg);
        v1 = live ? w1 : v1;
        v2 = live ? w2 : v2;
    }
    asm volatile("" : "+v"(v1), "+v"(v2));                               // (keeps the scheduler from moving the step into the dual phases)
    __builtin_amdgcn_sched_barrier(0);

    float wreg[kNlSlots], preg[kNlSlots];
    bool oks[kNlSlots];
    nf_dual_own<TY, BORDER, 0>(L, a, rs, pb, p_old, p_new, x, y, lx, ly, i4, in, wt, rwt, bp0, wreg, preg, oks);
    NfBatch bq0;
    nf_batch_loads<1>(rs, pb, q_old, 0, i4, bq0);                        // the second component's first batch: in flight across the
    __builtin_amdgcn_sched_barrier(0);                                   // crossing edges, two barriers and the first divergence
    if (!(kNfWhatIf & 2)) nf_cross_compute<TY>(L, a, X0, Y0, cr);
    __syncthreads();
    float dp = in ? nf_divergence<TY, BORDER>(L, lx, ly, wreg, preg, oks) : 0.0f;
    __syncthreads();
    nf_dual_own<TY, BORDER, 1>(L, a, rs, pb, q_old, q_new, x, y, lx, ly, i4, in, wt, rwt, bq0, wreg, preg, oks);
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
    if (kNfCheck) {                                // (every workgroup writes the same three values; the host zeroes g_nf_check)
        g_nf_extent[0] = 4u * (unsigned)(a.nx * a.ny);
        g_nf_extent[1] = 4u * (unsigned)a.pe;
        g_nf_extent[2] = 4u * (unsigned)a.pe * (unsigned)N_COUNT;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicAdd(&g_nf_check[3], 1u);
            if (a.check_selftest) {
                (void)L.pq_at(kNlSlots, 0, 0);
                (void)L.ub_at(0, TY + 2 * kNlR, 0);
                (void)L.im_at(1, 0, kNfW + 2 * kCsR);
                (void)L.pq_flat(kNlSlots * (TY + 2 * kNlR) * kNfLW);
                (void)nf_checked_off(4u * (unsigned)a.pe, 4u * (unsigned)(a.nx * a.ny));
            }
        }
    }
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
