// tvl1_persist_kernel.hpp -- the whole static TV-L1 schedule of ONE pair in ONE launch (gfx950 / MI355X).
//
// The per-launch kernel of tvl1_kernels.hpp pays ~8 us per launch besides its iterations (launch gap, loading and storing the
// whole tile state, the end-of-kernel write-back) plus one pointwise launch per outer warp; at 1024x436 with 10 iterations per
// warp (BASELINE configs[1]) that is a quarter of the solve.  When the pair's tiles fit the chip one-per-CU, this kernel keeps
// every tile's state in registers for the whole solve (all `warps` x `max_inner` iterations) and only exchanges the halo:
//
//   per outer warp     bicubic warp of the owned pixels (rho_c, I1wx, I1wy stay in registers)
//   per chunk of k     publish the owned pixels near the tile edge (write-through 16-B stores into the tile's exchange slab,
//   iterations         double buffered), raise the tile's epoch flag; wait for the (up to) eight neighbours' flags; fetch the
//                      halo from their slabs; k iterations (tvl1_iterate, the same code as the per-launch kernel)
//
// No grid-wide barrier: a tile only waits for its neighbours.  The arithmetic and its order per pixel are those of the
// per-launch path, and every halo value is the owner's value, so the results are bit-identical to it (tested).
//
// Inter-workgroup visibility (per-CU L1 and per-XCD L2 are not coherent): every handed-off byte is stored with sc1 (write-
// through), each storing wave drains (s_waitcnt vmcnt(0)) before the workgroup's barrier, one lane then stores the flag with
// an agent-scope atomic store; the consumer polls the flags with agent-scope atomic loads (one wave), a workgroup barrier, then
// loads the bytes with sc1 loads only (bypass L1; the lines are not in this XCD's L2 before the first load after the
// store, because sc1 stores drop them).  No handed-off byte is ever read by a plain load inside the launch.
//
// Residency: one workgroup per tile, tiles <= CUs (host-checked), plain launch, and the host only takes this path when nothing
// else of the process is on the device (faldoi_hip.hip, DeviceGate).  Every spin is bounded: after `spin_limit` polls a workgroup
// raises the abort word and returns; every poller also watches the abort word, so the grid drains and the host redoes the solve
// on the per-launch path instead of hanging.
#pragma once
#include "tvl1_kernels.hpp"

#pragma clang fp contract(off)

namespace faldoi {

constexpr int kXPlanes = 11;       // exchange slots: u1 u2 ub1 ub2 xi11 xi12 xi21 xi22 rho_c I1wx I1wy
constexpr int kFlagStride = 32;    // unsigned words between two tiles' flags: one 128-B line each
constexpr int kPersistStamps = 16;  // stamps per tile (measurement aid)
#ifndef FALDOI_PERSIST_EARLY_DUALS
#define FALDOI_PERSIST_EARLY_DUALS 1
#endif
constexpr bool kPersistEarlyDuals = FALDOI_PERSIST_EARLY_DUALS != 0;
#ifndef FALDOI_PERSIST_EARLY_FLOW
#define FALDOI_PERSIST_EARLY_FLOW 1
#endif
constexpr bool kPersistEarlyFlow = kPersistEarlyDuals && FALDOI_PERSIST_EARLY_FLOW != 0;

typedef unsigned int xvec4 __attribute__((ext_vector_type(4)));

struct PersistArgs {
    float *base;                   // the pair's slab (planes of tvl1_kernels.hpp)
    size_t plane_elems;
    int nx, ny, ntx, nty;
    int hx, hy;                    // halo columns (= iterations per chunk) and halo rows (multiple of R, >= hx)
    int yshift;                    // the tile grid starts this many rows ABOVE the image (a multiple of R): tile row j owns image rows
                                   // [j Sy - yshift, (j+1) Sy - yshift).  The rows a grid has to spare (nty Sy - ny) are split between the
                                   // first and the last tile row instead of all going to the last: both then hold about half a tile of
                                   // image rows, and the tiles that run the row / corner variants of the iteration (the slowest
                                   // instruction streams, profiles/r04_frame_tiles.md) stop pacing their neighbours
    TvConst prm;                   // prm.chunk = iterations per exchange
    float *err_out;                // [warps][ntiles][kMaxChunk]
    float *xslab;                  // [2][ntiles][kXPlanes][NW][64] x 16 B
    unsigned xslab_bytes;
    unsigned *flags;               // [ntiles] epoch flags, kFlagStride words apart; then the abort word
    unsigned spin_limit;
    unsigned long long *stamps;    // optional: [ntiles][kPersistStamps], taken during warp `stamp_warp`
    int stamp_warp;
    int *unsafe;                   // raised when a |grad I|^2 does not admit the one-product quotient (PairCtl::unsafe of the pair)
    // tolerance stop on (tvl1_persist_dyn_kernel.hpp) only:
    unsigned *err_board;           // [2][ntiles][kMaxChunk] per-iteration tile maxima of the chunk just iterated (float bits)
    int *iters_out;                // [warps] iterations each warp ran, as the reference's `Warping: w,Iter: n` line
    float *werr_out;               // [warps] its last max |du|^2
};

template <int R>
__device__ __forceinline__ xvec4 pack_rows(const float (&v)[R])
{
    xvec4 o = { 0u, 0u, 0u, 0u };
    o.x = __float_as_uint(v[0]);
    if (R > 1) o.y = __float_as_uint(v[R > 1 ? 1 : 0]);
    if (R > 2) o.z = __float_as_uint(v[R > 2 ? 2 : 0]);
    if (R > 3) o.w = __float_as_uint(v[R > 3 ? 3 : 0]);
    return o;
}
template <int R>
__device__ __forceinline__ void unpack_rows(float (&v)[R], const xvec4 o, bool take)
{
    v[0] = take ? __uint_as_float(o.x) : v[0];
    if (R > 1) v[R > 1 ? 1 : 0] = take ? __uint_as_float(o.y) : v[R > 1 ? 1 : 0];
    if (R > 2) v[R > 2 ? 2 : 0] = take ? __uint_as_float(o.z) : v[R > 2 ? 2 : 0];
    if (R > 3) v[R > 3 ? 3 : 0] = take ? __uint_as_float(o.w) : v[R > 3 ? 3 : 0];
}

// ---- pieces shared by the static body below and the tolerance-stop body (tvl1_persist_dyn_kernel.hpp) ----

// Where a thread sits: tile = 64 columns x R*NW rows, halo hx columns / hy rows; lane = column, wave w = rows [wR, wR+R).
struct PxGeom {
    int gx, gy0;                   // image column; first image row of the thread's R rows (wave-uniform, a multiple of R)
    bool col_in, wave_live;        // column inside the image; the wave has rows inside the image
    bool owner, halo, ring;        // its pixels are stored by this tile | by another tile | owned and inside a neighbour's footprint
    unsigned plane_stride, parity_stride;   // exchange slab addressing, in bytes (host: the whole slab is below 4 GB)
    unsigned pub_off, src_off;     // the thread's own vector in its tile's slab | the owner's vector of its halo pixels
};

template <int R, int NW>
__device__ __forceinline__ PxGeom px_geometry(const PersistArgs &a, int t, int txi, int tyi, int lane, int wv)
{
    constexpr int TH = R * NW;
    const int Sx = kTileW - 2 * a.hx, Sy = TH - 2 * a.hy;
    PxGeom g;
    g.gx = txi * Sx - a.hx + lane;
    const int own_lo = tyi * Sy - a.yshift, own_hi = own_lo + Sy;            // image rows this tile owns (clipped by the image below)
    g.gy0 = own_lo - a.hy + wv * R;                    // hy, Sy and yshift are multiples of R: a thread's rows share their owner
    g.col_in = g.gx >= 0 && g.gx < a.nx;
    const bool col_own = g.col_in && g.gx >= txi * Sx && g.gx < (txi + 1) * Sx;
    const bool group_in = g.gy0 >= 0 && g.gy0 < a.ny;
    const bool group_own = group_in && g.gy0 >= own_lo && g.gy0 < own_hi;
    g.wave_live = group_in;
    g.owner = col_own && group_own;
    g.halo = g.col_in && group_in && !g.owner;
    g.ring = g.owner && (g.gx - txi * Sx < a.hx || (txi + 1) * Sx - 1 - g.gx < a.hx || g.gy0 - own_lo < a.hy ||
                         own_hi - (g.gy0 + R) < a.hy);
    g.plane_stride = (unsigned)(NW * kTileW) * 16u;
    const unsigned tile_stride = g.plane_stride * (unsigned)kXPlanes;
    g.parity_stride = tile_stride * (unsigned)(a.ntx * a.nty);
    g.pub_off = (unsigned)t * tile_stride + (unsigned)(wv * kTileW + lane) * 16u;
    g.src_off = 0;
    if (g.halo) {
        const int txp = g.gx / Sx, typ = (g.gy0 + a.yshift) / Sy;
        const int lane_p = g.gx - txp * Sx + a.hx, wave_p = (g.gy0 + a.yshift - typ * Sy + a.hy) / R;
        g.src_off = (unsigned)(typ * a.ntx + txp) * tile_stride + (unsigned)(wave_p * kTileW + lane_p) * 16u;
    }
    return g;
}

// The uploaded state of the owned pixels (state buffer 0); everything else starts at zero.
template <int R>
__device__ __forceinline__ void px_load_owned(TileState<R> &s, const PersistArgs &a, const PxGeom &g)
{
    const float *ini = a.base + (size_t)P_STATE0 * a.plane_elems;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int gy = g.gy0 + r;
        s.own[r] = g.owner && gy < a.ny;
        const int p = s.own[r] ? gy * a.nx + g.gx : 0;
        s.u1[r] = s.own[r] ? ini[0 * a.plane_elems + p] : 0.0f;
        s.u2[r] = s.own[r] ? ini[1 * a.plane_elems + p] : 0.0f;
        s.x11[r] = s.own[r] ? ini[4 * a.plane_elems + p] : 0.0f;
        s.x12[r] = s.own[r] ? ini[5 * a.plane_elems + p] : 0.0f;
        s.x21[r] = s.own[r] ? ini[6 * a.plane_elems + p] : 0.0f;
        s.x22[r] = s.own[r] ? ini[7 * a.plane_elems + p] : 0.0f;
        s.ub1[r] = s.ub2[r] = 0.0f;
        s.rc[r] = s.ix[r] = s.iy[r] = s.gr[r] = 0.0f;
        s.rgr[r] = 0.0;
    }
}

// The solved state of the owned pixels into state buffer 1 (ubar is dead after the last warp).
template <int R>
__device__ __forceinline__ void px_store_owned(const TileState<R> &s, const PersistArgs &a, const PxGeom &g)
{
    float *dst = a.base + (size_t)P_STATE1 * a.plane_elems;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (s.own[r]) {
            const int p = (g.gy0 + r) * a.nx + g.gx;
            dst[0 * a.plane_elems + p] = s.u1[r];
            dst[1 * a.plane_elems + p] = s.u2[r];
            dst[4 * a.plane_elems + p] = s.x11[r];
            dst[5 * a.plane_elems + p] = s.x12[r];
            dst[6 * a.plane_elems + p] = s.x21[r];
            dst[7 * a.plane_elems + p] = s.x22[r];
        }
    }
}


// The warp constants of the thread's owned pixels (src/global_faldoi.cpp:635-657; the code of tvl1_warp_kernel).
// (Dealing the owned pixels to all threads through LDS, two pixels each instead of R for the owners and none for the halo, was
// measured: no faster -- the phase is bound by its ~430 vector instructions per pixel, not by their spread.)
template <int R>
__device__ __forceinline__ void px_warp_constants(TileState<R> &s, const float *__restrict__ I0, const float *__restrict__ I1,
                                                  const float *__restrict__ I1x, const float *__restrict__ I1y, int gx, int gy0,
                                                  int nx, int ny)
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float rc = 0.0f, wx = 0.0f, wy = 0.0f;                   // the halo's constants arrive with the exchange
        if (!(kWhatIf & 32) && s.own[r]) {
            const int gy = gy0 + r;
            float i0v = I0[gy * nx + gx];
            asm volatile("" : "+v"(i0v));
            const float uu = (float)gx + s.u1[r];                // src/bicubic_interpolation.c:259-260
            const float vv = (float)gy + s.u2[r];
            const BicubicTaps tp = bicubic_taps(uu, vv, nx, ny);
            float w0 = 0.0f;                                     // border_out = true (:635-637)
            if (!tp.clamped) {
                w0 = bicubic_eval_rows(I1, tp, nx);
                wx = bicubic_eval_rows(I1x, tp, nx);
                wy = bicubic_eval_rows(I1y, tp, nx);
            }
            rc = rho_const(w0, wx, wy, s.u1[r], s.u2[r], i0v);
        }
        s.rc[r] = rc;
        s.ix[r] = wx;
        s.iy[r] = wy;
    }
}

// Publish: write-through, one 16-byte vector (the thread's R rows) per plane.  `first`: the exchange at a warp's start carries the
// warp constants and no ubar (ubar = u there, :659-660).  Called by the threads of the ring only.
template <int R, typename RSRC>
__device__ __forceinline__ void px_publish_duals(const TileState<R> &s, RSRC rsrc, unsigned base, unsigned plane_stride)
{
    __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.x11), rsrc, base + 4u * plane_stride, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.x12), rsrc, base + 5u * plane_stride, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.x21), rsrc, base + 6u * plane_stride, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.x22), rsrc, base + 7u * plane_stride, 0, 16);
}
template <int R, typename RSRC>
__device__ __forceinline__ void px_publish_flow(const TileState<R> &s, RSRC rsrc, unsigned base, unsigned plane_stride)
{
    __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.u1), rsrc, base + 0u * plane_stride, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.u2), rsrc, base + 1u * plane_stride, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.ub1), rsrc, base + 2u * plane_stride, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.ub2), rsrc, base + 3u * plane_stride, 0, 16);
}
// `duals` / `flow`: false when px_publish_duals / px_publish_flow has stored them already (from inside the previous chunk's last iteration)
template <int R, typename RSRC>
__device__ __forceinline__ void px_publish(const TileState<R> &s, RSRC rsrc, unsigned base, unsigned plane_stride, bool first,
                                           bool duals = true, bool flow = true)
{
    if (flow) {
        __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.u1), rsrc, base + 0u * plane_stride, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.u2), rsrc, base + 1u * plane_stride, 0, 16);
    }
    if (flow && !first) {
        __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.ub1), rsrc, base + 2u * plane_stride, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.ub2), rsrc, base + 3u * plane_stride, 0, 16);
    }
    if (duals) px_publish_duals<R>(s, rsrc, base, plane_stride);
    if (first) {
        __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.rc), rsrc, base + 8u * plane_stride, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.ix), rsrc, base + 9u * plane_stride, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(pack_rows<R>(s.iy), rsrc, base + 10u * plane_stride, 0, 16);
    }
}

// Fetch the halo from its owners' slabs (sc1 loads only); at a warp's start also ubar = u and the per-warp |grad I|^2.
template <int R, bool ONCE, typename RSRC>
__device__ __forceinline__ void px_fetch(TileState<R> &s, RSRC rsrc, unsigned base, unsigned plane_stride, bool halo, bool first,
                                         int *unsafe)
{
    xvec4 q[kXPlanes];
#pragma unroll
    for (int k = 0; k < kXPlanes; ++k) q[k] = xvec4{ 0u, 0u, 0u, 0u };
    if (halo) {
        q[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 0u * plane_stride, 0, 16);
        q[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 1u * plane_stride, 0, 16);
        if (!first) {
            q[2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 2u * plane_stride, 0, 16);
            q[3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 3u * plane_stride, 0, 16);
        }
        q[4] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 4u * plane_stride, 0, 16);
        q[5] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 5u * plane_stride, 0, 16);
        q[6] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 6u * plane_stride, 0, 16);
        q[7] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 7u * plane_stride, 0, 16);
        if (first) {
            q[8] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 8u * plane_stride, 0, 16);
            q[9] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 9u * plane_stride, 0, 16);
            q[10] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 10u * plane_stride, 0, 16);
        }
    }
    unpack_rows<R>(s.u1, q[0], halo);
    unpack_rows<R>(s.u2, q[1], halo);
    unpack_rows<R>(s.ub1, q[2], halo && !first);
    unpack_rows<R>(s.ub2, q[3], halo && !first);
    unpack_rows<R>(s.x11, q[4], halo);
    unpack_rows<R>(s.x12, q[5], halo);
    unpack_rows<R>(s.x21, q[6], halo);
    unpack_rows<R>(s.x22, q[7], halo);
    unpack_rows<R>(s.rc, q[8], halo && first);
    unpack_rows<R>(s.ix, q[9], halo && first);
    unpack_rows<R>(s.iy, q[10], halo && first);
    if (first) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s.ub1[r] = s.u1[r];                                    // start of a warp: ubar = u (:659-660)
            s.ub2[r] = s.u2[r];
            s.gr[r] = grad_sq(s.ix[r], s.iy[r]);                   // :648-652
            s.rgr[r] = ONCE ? recip_f64_full(s.gr[r]) : recip_f64(s.gr[r]);
        }
        // every lane of the wave is here (`first` is uniform); a divisor that needs the general quotient is reported to the host
        if (ONCE && !tile_div_once<R>(s) && (threadIdx.x & 63) == 0) *unsafe = 1;
    }
}

// One wave waits until every neighbour (lanes 0..7 one each; `all`: every tile of the grid, lane by lane) has published `epoch`;
// lane 8 (`all`: an extra load by lane 0) watches the abort word.  Bounded: returns true when the workgroup must leave.
__device__ __forceinline__ bool px_wait(const unsigned *flags, unsigned *abort_word, unsigned epoch, int lane, int t, int txi,
                                        int tyi, int ntx, int nty, unsigned spin_limit, bool all)
{
    const int ntiles = ntx * nty;
    const int d = lane < 4 ? lane : lane + 1;                     // 0..8 without the centre
    const int ax = txi + d % 3 - 1, ay = tyi + d / 3 - 1;
    const bool has = lane < 8 && ax >= 0 && ax < ntx && ay >= 0 && ay < nty;
    const unsigned *pw = (lane == 8) ? abort_word : flags + (size_t)(has ? ay * ntx + ax : t) * kFlagStride;
    if (spin_limit == 0u) {                                       // test hook (FALDOI_PERSIST_SPINS=0): give up before the first poll
        if (lane == 0) __hip_atomic_store(abort_word, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    for (unsigned spins = 0;; ++spins) {
        bool behind = false, aborted = false;
        if (!all) {
            unsigned v = 0;
            if (lane < 9) v = __hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            behind = !(kWhatIf & 16) && has && (int)(v - epoch) < 0;
            aborted = lane == 8 && v != 0u;
        } else {                                                  // host: ntiles <= 256; the four loads of a lane go out together
            unsigned v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = lane + 64 * j;
                v[j] = __hip_atomic_load(flags + (size_t)(q < ntiles ? q : t) * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned ab = lane == 0 ? __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            int lag = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) lag |= (int)(v[j] - epoch);   // negative if any of them is behind
            behind = lag < 0;
            aborted = ab != 0u;
        }
        if (__builtin_amdgcn_ballot_w64(aborted) != 0ull) return true;
        if (__builtin_amdgcn_ballot_w64(behind) == 0ull) return false;
        if (spins >= spin_limit) {
            if (lane == 0) __hip_atomic_store(abort_word, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

template <int R, int NW, int BORDER, bool ONCE>
__device__ __forceinline__ void tvl1_persist_body(const PersistArgs &a, int t, int txi, int tyi,
                                                  float (*s_ub1)[kTileW], float (*s_ub2)[kTileW], float (*s_x12)[kTileW],
                                                  float (*s_x22)[kTileW], unsigned (*s_err)[kTileW], int *s_ctl)
{
    static_assert(R >= 2 && R <= 4, "a thread's rows travel as one 16-byte vector");
    const int nx = a.nx, ny = a.ny, ntx = a.ntx, nty = a.nty;
    const int ntiles = ntx * nty;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const PxGeom g = px_geometry<R, NW>(a, t, txi, tyi, lane, wv);
    const int gx = g.gx, gy0 = g.gy0;
    const bool wave_live = g.wave_live, halo = g.halo, ring = g.ring;
    const unsigned plane_stride = g.plane_stride, parity_stride = g.parity_stride, pub_off = g.pub_off, src_off = g.src_off;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xslab, 0, a.xslab_bytes, 0x00020000);
    unsigned *abort_word = a.flags + (size_t)ntiles * kFlagStride;
    const float *I0 = a.base + (size_t)P_I0 * a.plane_elems;
    const float *I1 = a.base + (size_t)P_I1 * a.plane_elems;
    const float *I1x = a.base + (size_t)P_I1X * a.plane_elems;
    const float *I1y = a.base + (size_t)P_I1Y * a.plane_elems;

    TileState<R> s;
    px_load_owned<R>(s, a, g);
    for (int i = wv; i < kMaxChunk; i += NW) s_err[i][lane] = 0u;

    unsigned epoch = 0;
    for (int w = 0; w < a.prm.warps; ++w) {
        const bool stamp = a.stamps && w == a.stamp_warp && threadIdx.x == 0;
        unsigned long long *st = a.stamps + (size_t)t * kPersistStamps;
        if (stamp) { st[0] = __builtin_amdgcn_s_memrealtime(); st[13] = __builtin_amdgcn_s_memtime(); }
        px_warp_constants<R>(s, I0, I1, I1x, I1y, gx, gy0, nx, ny);
        if (stamp) {
            float sink = s.rc[0] + s.rc[R - 1] + s.iy[R - 1];
            asm volatile("" : "+v"(sink));
            st[1] = __builtin_amdgcn_s_memrealtime();
        }

        for (int n = 0; n < a.prm.max_inner;) {
            const int left = a.prm.max_inner - n;
            const int it = left < a.prm.chunk ? left : a.prm.chunk;
            const bool first = n == 0;
            ++epoch;
            const unsigned par = (epoch & 1u) * parity_stride;

            // (the ring's duals of every exchange but the first of the launch went out from inside the previous chunk's last iteration:
            // they are final after its dual step, and half of the exchange's bytes drain while the primal step computes)
            // (publishing the flow early as well at a warp's end -- it then drains during the warp constants -- was measured: no
            // further gain, 11.93 against 11.90 ms: the exchange at a warp's start waits for the owners' bicubic phase, not for bytes)
            if (ring) px_publish<R>(s, rsrc, par + pub_off, plane_stride, first, !(kPersistEarlyDuals && epoch > 1),
                                    !(kPersistEarlyFlow && !first));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains before the barrier
            if (a.stamps && w == a.stamp_warp && threadIdx.x == 64 * (NW / 2)) st[first ? 7 : 12] = __builtin_amdgcn_s_memrealtime();
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_store(a.flags + (size_t)t * kFlagStride, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (stamp) st[first ? 2 : 8] = __builtin_amdgcn_s_memrealtime();
            }

            if (wv == 0) {                                        // one wave waits for the neighbours' epoch
                const bool failed = px_wait(a.flags, abort_word, epoch, lane, t, txi, tyi, ntx, nty, a.spin_limit, false);
                if (lane == 0) s_ctl[0] = failed ? 1 : 0;
            }
            __syncthreads();
            if (s_ctl[0]) return;                                 // uniform: the whole workgroup leaves
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the loads below the poll
            if (stamp) st[first ? 3 : 9] = __builtin_amdgcn_s_memrealtime();

            px_fetch<R, ONCE>(s, rsrc, par + src_off, plane_stride, halo, first, a.unsafe);
            if (stamp) {
                float sink = s.u1[0] + s.x22[R - 1] + s.iy[R - 1];
                asm volatile("" : "+v"(sink));
                st[first ? 4 : 10] = __builtin_amdgcn_s_memrealtime();
            }
            s_ub1[wv][lane] = s.ub1[0];
            s_ub2[wv][lane] = s.ub2[0];

            const bool ends_warp = n + it >= a.prm.max_inner;
            // the slab the NEXT exchange uses (other parity): every neighbour has finished reading it -- it needed this tile's
            // flag of the CURRENT exchange to get where it is, and raised its own only after its previous fetch
            const bool more = !(ends_warp && w == a.prm.warps - 1);
            const unsigned par_next = ((epoch + 1u) & 1u) * parity_stride;
            auto early_duals = [&](int iter) {
                if (kPersistEarlyDuals && more && ring && iter == it - 1) px_publish_duals<R>(s, rsrc, par_next + pub_off, plane_stride);
            };
            // and in a chunk that does not end its warp, flow and ubar go the same way once the last iteration's primal step is done:
            // before the workgroup's end-of-chunk barrier instead of behind it
            auto early_flow = [&](int iter) {
                if (kPersistEarlyFlow && !ends_warp && ring && iter == it - 1) px_publish_flow<R>(s, rsrc, par_next + pub_off, plane_stride);
            };
            tvl1_iterate<R, NW, BORDER, ONCE>(s, it, a.hy, gx, gy0, nx, ny, lane, wv, wave_live, a.prm, s_ub1, s_ub2, s_x12, s_x22,
                                        s_err, early_duals, early_flow);
            if (stamp) st[first ? 5 : 11] = __builtin_amdgcn_s_memrealtime();

            __syncthreads();                                       // the last iteration's maxima are in s_err
            if (ends_warp)                                         // the chunk that ends the warp: what the stats kernel reads
                tvl1_store_errors<NW>(s_err, it, lane, wv, a.err_out + ((size_t)w * ntiles + t) * kMaxChunk);
            for (int i = wv; i < kMaxChunk; i += NW) s_err[i][lane] = 0u;   // the same wave that read row i
            n += it;
        }
        if (stamp) { st[6] = __builtin_amdgcn_s_memrealtime(); st[14] = __builtin_amdgcn_s_memtime(); }
    }

    px_store_owned<R>(s, a, g);
}

template <int R, int NW, bool ONCE>
__global__ __launch_bounds__(64 * NW) void tvl1_persist_kernel(PersistArgs a)
{
    constexpr int TH = R * NW;
    __shared__ float s_ub1[NW][kTileW];
    __shared__ float s_ub2[NW][kTileW];
    __shared__ float s_x12[NW][kTileW];
    __shared__ float s_x22[NW][kTileW];
    __shared__ unsigned s_err[kMaxChunk][kTileW];
    __shared__ int s_ctl[4];

    const int per = gridDim.x >> 3;                                // XCD-aware tile order, as tvl1_chunk_kernel
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= a.ntx * a.nty) return;
    const int tyi = t / a.ntx, txi = t - tyi * a.ntx;
    const int X0 = txi * (kTileW - 2 * a.hx) - a.hx, Y0 = tyi * (TH - 2 * a.hy) - a.hy - a.yshift;
    const bool colb = X0 <= 0 || X0 + kTileW >= a.nx, rowb = Y0 <= 0 || Y0 + TH >= a.ny;
#define FALDOI_BODY(B) tvl1_persist_body<R, NW, B, ONCE>(a, t, txi, tyi, s_ub1, s_ub2, s_x12, s_x22, s_err, s_ctl)
    if (colb && rowb) FALDOI_BODY(3);
    else if (colb) FALDOI_BODY(1);
    else if (rowb) FALDOI_BODY(2);
    else FALDOI_BODY(0);
#undef FALDOI_BODY
}

}  // namespace faldoi
