// tvl1_persist_dyn_kernel.hpp -- the persistent TV-L1 kernel with the reference's tolerance stop (gfx950 / MI355X).
//
// The reference ends a warp when max |du|^2 of an iteration is at most tol^2 (`while (err_D > tol*tol && n < MAX_ITERATIONS_GLOBAL)`,
// src/global_faldoi.cpp:684) -- a maximum over the WHOLE image, once per iteration.  The launch-per-chunk path decides it in a
// control kernel between two launches (tvl1_ctl_kernel); here, with every tile resident in registers for the whole solve
// (tvl1_persist_kernel.hpp), the tiles decide it together at the end of every chunk of k iterations:
//
//   per outer warp   warp constants; exchange with the neighbours (as the static kernel)
//   per chunk        snapshot of the tile's state in LDS; k iterations; publish the ring AND the chunk's k tile maxima (error board,
//                    write-through), raise the epoch flag; wait for EVERY tile's flag; every workgroup reduces the board to the k
//                    image maxima and takes the same decision as tvl1_ctl_kernel:
//                      no iteration passed the test          -> fetch the halo, next chunk
//                      the last iteration of the chunk did   -> the warp ends with the state as it is
//                      an earlier one did (iteration i)      -> restore the snapshot and iterate exactly i + 1 times: the state the
//                                                               reference stops in; the warp ends
// The ring published at the end of a warp is never fetched (the next warp starts with an exchange of its own), and the all-tiles
// wait is the same bounded spin as the neighbours' wait, with the same abort word.  Iteration counts and errors per warp are
// written by tile 0 and equal the reference's `Warping: w,Iter: n Error: e` lines (tests: every tolerance-stop test of the suite
// runs through this kernel when the pair fits the chip).
#pragma once
#include "tvl1_persist_kernel.hpp"

#pragma clang fp contract(off)

namespace faldoi {

template <int R, int NW, int BORDER, bool ONCE>
__device__ __forceinline__ void tvl1_persist_dyn_body(const PersistArgs &a, int t, int txi, int tyi,
                                                      float (*s_ub1)[kTileW], float (*s_ub2)[kTileW], float (*s_x12)[kTileW],
                                                      float (*s_x22)[kTileW], unsigned (*s_err)[kTileW], int *s_ctl, float *s_dec,
                                                      float *s_snap)
{
    static_assert(R >= 2 && R <= 4, "a thread's rows travel as one 16-byte vector");
    constexpr int NT = 64 * NW;
    const int nx = a.nx, ny = a.ny, ntx = a.ntx, nty = a.nty;
    const int ntiles = ntx * nty;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const PxGeom g = px_geometry<R, NW>(a, t, txi, tyi, lane, wv);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xslab, 0, a.xslab_bytes, 0x00020000);
    unsigned *abort_word = a.flags + (size_t)ntiles * kFlagStride;
    const float *I0 = a.base + (size_t)P_I0 * a.plane_elems;
    const float *I1 = a.base + (size_t)P_I1 * a.plane_elems;
    const float *I1x = a.base + (size_t)P_I1X * a.plane_elems;
    const float *I1y = a.base + (size_t)P_I1Y * a.plane_elems;

    TileState<R> s;
    px_load_owned<R>(s, a, g);
    for (int i = wv; i < kMaxChunk; i += NW) s_err[i][lane] = 0u;

    unsigned epoch = 0, cseq = 0;                                  // exchanges so far; chunk ends so far
    for (int w = 0; w < a.prm.warps; ++w) {
        px_warp_constants<R>(s, I0, I1, I1x, I1y, g.gx, g.gy0, nx, ny);

        // ---- the exchange that starts a warp: flow, duals, warp constants; neighbours only ----
        ++epoch;
        {
            const unsigned par = (epoch & 1u) * g.parity_stride;
            if (g.ring) px_publish<R>(s, rsrc, par + g.pub_off, g.plane_stride, true);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.flags + (size_t)t * kFlagStride, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wv == 0) {
                const bool failed = px_wait(a.flags, abort_word, epoch, lane, t, txi, tyi, ntx, nty, a.spin_limit, false);
                if (lane == 0) s_ctl[0] = failed ? 1 : 0;
            }
            __syncthreads();
            if (s_ctl[0]) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            px_fetch<R, ONCE>(s, rsrc, par + g.src_off, g.plane_stride, g.halo, true, a.unsafe);
        }

        int n = 0;
        float e_last = 0.0f;
        for (;;) {
            const int left = a.prm.max_inner - n;
            const int it = left < a.prm.chunk ? left : a.prm.chunk;

            // ---- snapshot: what an exact-length re-run starts from ----
#pragma unroll
            for (int r = 0; r < R; ++r) {
                s_snap[(0 * R + r) * NT + tid] = s.u1[r];
                s_snap[(1 * R + r) * NT + tid] = s.u2[r];
                s_snap[(2 * R + r) * NT + tid] = s.ub1[r];
                s_snap[(3 * R + r) * NT + tid] = s.ub2[r];
                s_snap[(4 * R + r) * NT + tid] = s.x11[r];
                s_snap[(5 * R + r) * NT + tid] = s.x12[r];
                s_snap[(6 * R + r) * NT + tid] = s.x21[r];
                s_snap[(7 * R + r) * NT + tid] = s.x22[r];
            }
            s_ub1[wv][lane] = s.ub1[0];
            s_ub2[wv][lane] = s.ub2[0];
            // The ring leaves from inside the chunk's last iteration, as in the static kernel (tvl1_persist_kernel.hpp): the duals
            // after its dual step, flow and ubar after its primal step, into the slab of the exchange that follows (every neighbour
            // is done reading that slab: it published the current exchange only after its previous fetch).  Published on the
            // assumption that the warp goes on, like the whole ring before; if the stop test ends the warp instead -- also after an
            // exact-length re-run, which publishes nothing -- nobody fetches it, and the next warp's first exchange writes every
            // plane it reads into the other slab.
            const unsigned par_next = ((epoch + 1u) & 1u) * g.parity_stride;
            auto early_duals = [&](int iter) {
                if (kPersistEarlyDuals && g.ring && iter == it - 1) px_publish_duals<R>(s, rsrc, par_next + g.pub_off, g.plane_stride);
            };
            auto early_flow = [&](int iter) {
                if (kPersistEarlyDuals && g.ring && iter == it - 1) px_publish_flow<R>(s, rsrc, par_next + g.pub_off, g.plane_stride);
            };
            tvl1_iterate<R, NW, BORDER, ONCE>(s, it, a.hy, g.gx, g.gy0, nx, ny, lane, wv, g.wave_live, a.prm, s_ub1, s_ub2, s_x12, s_x22,
                                        s_err, early_duals, early_flow);
            __syncthreads();                                       // every iteration's per-lane maxima are in s_err

            // ---- publish the ring and this chunk's tile maxima; wait for every tile ----
            ++epoch;
            const unsigned par = (epoch & 1u) * g.parity_stride;
            // Two boards, alternating per chunk end: between a tile reading board c and another writing board c + 1 there is no
            // all-tiles wait (a warp-start exchange only couples neighbours), but nobody writes board c + 2 before everyone has
            // published c + 1, i.e. has left board c behind.  (The ring slabs alternate per exchange: only neighbours read them.)
            unsigned *board = a.err_board + (size_t)(cseq & 1u) * ntiles * kMaxChunk;
            ++cseq;
            for (int i = wv; i < kMaxChunk; i += NW) {             // wave i reduces iteration i's 64 lanes
                float m = (i < it) ? __uint_as_float(s_err[i][lane]) : 0.0f;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float other = __shfl_xor(m, o);
                    m = other > m ? other : m;
                }
                if (lane == 0) __hip_atomic_store(board + (size_t)t * kMaxChunk + i, __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_err[i][lane] = 0u;
            }
            if (g.ring && !kPersistEarlyDuals) px_publish<R>(s, rsrc, par + g.pub_off, g.plane_stride, false);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ring and board entries drained by every storing wave
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.flags + (size_t)t * kFlagStride, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wv == 0) {
                const bool failed = px_wait(a.flags, abort_word, epoch, lane, t, txi, tyi, ntx, nty, a.spin_limit, true);
                int stop_at = -1;
                if (!failed) {
                    // e[i] = max over all tiles of their maxima (values >= 0; the maximum is exact and order-independent).
                    // A tile's 16 entries are 64 bytes: four 16-byte sc1 loads per tile, all tiles of a lane requested before
                    // the first one is used (one round trip, not one per entry).
                    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc(board, 0, (unsigned)(ntiles * kMaxChunk * 4), 0x00020000);
                    constexpr int kTilesPerLane = 4;               // host: ntiles <= 256
                    constexpr int kVec = 2;                        // host: at most 8 iterations per chunk on this kernel
                    xvec4 bq[kTilesPerLane][kVec];
#pragma unroll
                    for (int j = 0; j < kTilesPerLane; ++j) {
                        const int q = lane + 64 * j;
#pragma unroll
                        for (int v = 0; v < kVec; ++v) {
                            bq[j][v] = xvec4{ 0u, 0u, 0u, 0u };
                            if (q < ntiles && 4 * v < it)
                                bq[j][v] = __builtin_amdgcn_raw_buffer_load_b128(brsrc, (unsigned)(q * kMaxChunk * 4 + 16 * v), 0, 16);
                        }
                    }
                    float e[kMaxChunk];
#pragma unroll
                    for (int i = 4 * kVec; i < kMaxChunk; ++i) e[i] = 0.0f;
#pragma unroll
                    for (int v = 0; v < kVec; ++v) {
                        float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;
#pragma unroll
                        for (int j = 0; j < kTilesPerLane; ++j) {
                            const float a0 = __uint_as_float(bq[j][v].x), a1 = __uint_as_float(bq[j][v].y);
                            const float a2 = __uint_as_float(bq[j][v].z), a3 = __uint_as_float(bq[j][v].w);
                            m0 = a0 > m0 ? a0 : m0; m1 = a1 > m1 ? a1 : m1; m2 = a2 > m2 ? a2 : m2; m3 = a3 > m3 ? a3 : m3;
                        }
                        e[4 * v + 0] = m0; e[4 * v + 1] = m1; e[4 * v + 2] = m2; e[4 * v + 3] = m3;
                    }
#pragma unroll
                    for (int i = 0; i < kMaxChunk; ++i) {
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            const float other = __shfl_xor(e[i], o);
                            e[i] = other > e[i] ? other : e[i];
                        }
                    }
                    // the first iteration of this chunk after which the reference's while-condition fails (tvl1_ctl_kernel)
#pragma unroll
                    for (int i = kMaxChunk - 1; i >= 0; --i)
                        if (i < it && !(e[i] > a.prm.tol2)) stop_at = i;
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < kMaxChunk; ++i) s_dec[i] = e[i];
                    }
                }
                if (lane == 0) { s_ctl[0] = failed ? 1 : 0; s_ctl[1] = stop_at; }
            }
            __syncthreads();
            if (s_ctl[0]) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int stop_at = s_ctl[1];
            if (stop_at >= 0 && stop_at < it - 1) {
                // stopped inside the chunk: the same input, exactly stop_at + 1 iterations
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    s.u1[r] = s_snap[(0 * R + r) * NT + tid];
                    s.u2[r] = s_snap[(1 * R + r) * NT + tid];
                    s.ub1[r] = s_snap[(2 * R + r) * NT + tid];
                    s.ub2[r] = s_snap[(3 * R + r) * NT + tid];
                    s.x11[r] = s_snap[(4 * R + r) * NT + tid];
                    s.x12[r] = s_snap[(5 * R + r) * NT + tid];
                    s.x21[r] = s_snap[(6 * R + r) * NT + tid];
                    s.x22[r] = s_snap[(7 * R + r) * NT + tid];
                }
                s_ub1[wv][lane] = s.ub1[0];
                s_ub2[wv][lane] = s.ub2[0];
                tvl1_iterate<R, NW, BORDER, ONCE>(s, stop_at + 1, a.hy, g.gx, g.gy0, nx, ny, lane, wv, g.wave_live, a.prm, s_ub1, s_ub2, s_x12,
                                            s_x22, s_err);
                __syncthreads();
                for (int i = wv; i < kMaxChunk; i += NW) s_err[i][lane] = 0u;
                n += stop_at + 1;
                e_last = s_dec[stop_at];
                break;
            }
            n += it;
            e_last = s_dec[it - 1];
            if (stop_at >= 0 || n >= a.prm.max_inner) break;
            px_fetch<R, ONCE>(s, rsrc, par + g.src_off, g.plane_stride, g.halo, false, a.unsafe);
        }
        if (t == 0 && tid == 0) {
            a.iters_out[w] = n;
            a.werr_out[w] = e_last;
        }
        __syncthreads();                                           // s_dec / s_ctl are rewritten by the next warp's first chunk
    }
    px_store_owned<R>(s, a, g);
}

template <int R, int NW, bool ONCE>
__global__ __launch_bounds__(64 * NW) void tvl1_persist_dyn_kernel(PersistArgs a)
{
    constexpr int TH = R * NW;
    __shared__ float s_ub1[NW][kTileW];
    __shared__ float s_ub2[NW][kTileW];
    __shared__ float s_x12[NW][kTileW];
    __shared__ float s_x22[NW][kTileW];
    __shared__ unsigned s_err[kMaxChunk][kTileW];
    __shared__ int s_ctl[4];
    __shared__ float s_dec[kMaxChunk];
    extern __shared__ __attribute__((aligned(16))) float s_snap[];   // 8 planes x R rows x 64*NW threads (host-sized)

    const int per = gridDim.x >> 3;                                // XCD-aware tile order, as tvl1_chunk_kernel
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= a.ntx * a.nty) return;
    const int tyi = t / a.ntx, txi = t - tyi * a.ntx;
    const int X0 = txi * (kTileW - 2 * a.hx) - a.hx, Y0 = tyi * (TH - 2 * a.hy) - a.hy - a.yshift;
    const bool colb = X0 <= 0 || X0 + kTileW >= a.nx, rowb = Y0 <= 0 || Y0 + TH >= a.ny;
#define FALDOI_BODY(B) tvl1_persist_dyn_body<R, NW, B, ONCE>(a, t, txi, tyi, s_ub1, s_ub2, s_x12, s_x22, s_err, s_ctl, s_dec, s_snap)
    if (colb && rowb) FALDOI_BODY(3);
    else if (colb) FALDOI_BODY(1);
    else if (rowb) FALDOI_BODY(2);
    else FALDOI_BODY(0);
#undef FALDOI_BODY
}

}  // namespace faldoi
