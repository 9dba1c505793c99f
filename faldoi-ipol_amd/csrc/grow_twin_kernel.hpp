// grow_twin_kernel.hpp -- one growing, several windows at a time where the pop order allows it (round 4).
//
// A growing is sequential: pop the best candidate, fix its pixel, fill and minimise the window around it, push its neighbours, pop
// again (src/local_faldoi.cpp:891-937, :679-727).  One workgroup runs one growing, and a lone wavefront issues one instruction per
// ~5 cycles: a window costs ~53 k cycles of which ~41 k are the fill and the minimisation (profiles/r04_growing_detail.md).
//
// What CAN run side by side without changing a bit.  After candidate A has been popped, the queue already holds the candidates that
// come next unless A's own pushes overtake them.  The first wavefront loads the queue's three top levels (seven elements, with their
// pixels' fixed flags: one round trip) and walks them the way a heap hands out its elements in order -- a frontier that starts at the
// root and grows by the children of whatever it takes -- writing down the EXPECTED next pops: a stale candidate (its pixel is fixed, or
// an earlier guess of this step fixes it: the sequential loop pops and skips it, :912-916) or a window.  A window guess stands if its
// window overlaps neither A's nor an earlier guess's (|di| > 2 wr or |dj| > 2 wr); the first one that overlaps ends the list -- it has
// to see what the earlier windows write.  Up to K-1 guesses: slot 0 of the workgroup (threads 0..127) computes A's window, slot g the
// g-th guess's, at the same time, from the same state; the guesses' results are HELD BACK.
//
// Then the sequential order is replayed on the real queue.  A's candidates are pushed exactly as before.  For every expected pop in
// turn: is the queue's top this very element, bit for bit?  A stale one is popped (it is stale whatever happens); a window guess is
// popped, its results stored and its candidates pushed -- and only then is the next expectation tested, against the queue those
// pushes have changed.  The first mismatch ends the replay.  Nothing of a guess exists outside the workgroup until it is confirmed:
// its pixel's fix (fixed / out / ene, :926-932) lives in LDS, where the guess's own window reads it, and is written to the planes
// together with the window's results.  The heap therefore sees exactly the sequential operations in the sequential order;
// a guess that was wrong costs nothing but the idle lanes it occupied.  K = 2 serves 26 % of the windows this way on the final/hard
// pair (one whole-image growing 42.2 -> 34.9 s), K = 4 more.
//
// All slots execute the same workgroup barriers: the window functions take the slot's thread index and an `on` flag
// (harmonic_fill_window, window_solve<true> in grow_kernels.hpp), and the iteration loop of window_solve runs while any slot goes on.
// TV-l2 coupled functionals (-m 0 / 1), no image partitions (their out-of-window divergence statements would break the disjointness).
#pragma once
#include "grow_kernels.hpp"

#pragma clang fp contract(off)

namespace faldoi {

constexpr int kTwinPeek = 7;          // queue elements looked at: the heap's three top levels
constexpr int kTwinSeq = 7;           // expected pops written down, at most

template <int K> struct TwinShared {
    GrowShared h[K];
    int ctl[8 + 4 * kTwinMax];        // 0 have A | 1 guesses | 2 queue size at exit | 8+4s.. per slot: pixel i, j | commit | energy bits
    int filled[kTwinMax];
    int untrusted[2 * kTwinMax];      // per wavefront: its part of its slot's window has a pixel that is not trusted
    float cand[kTwinMax][4];          // a guess's candidate: sim, u, v (its pixel's fix, written to the planes only when confirmed)
};

// The four neighbours' state of a pixel and its own energy, read by wave 0 ahead of the window computation (insert_candidates
// :497-537 needs it): lane q < 4 holds neighbour q's, lane 4 the centre's -- one set of loads for all five
struct TwinNeigh {
    int nfix;
    float nene, nsal;
};
__device__ __forceinline__ void twin_neigh_load(const GrowDir &D, int nx, int ny, int i, int j, TwinNeigh &n)
{
    const int q = threadIdx.x & 63;
    const int px = i + (q == 2 ? 1 : q == 3 ? -1 : 0), py = j + (q == 0 ? 1 : q == 1 ? -1 : 0);     // insert_candidates :501-505
    const bool inside = px >= 0 && px < nx && py >= 0 && py < ny;
    const int g = inside ? py * nx + px : j * nx + i;
    const int f = D.fixed[g];
    n.nfix = inside ? f : 1;
    n.nene = D.ene[g];
    n.nsal = D.sal ? D.sal[g] : 1.0f;
}
// insert_candidates (:497-537) + the centre's update (:718-726) for one window, by wave 0; sm = the LDS block of the slot that
// minimised it (flow of the window in sm.a / sm.b), ener0 = its energy (wave-uniform)
__device__ __forceinline__ void twin_push(const GrowDir &D, GrowShared &sm, size_t pe, int nx, int ny, int i, int j, int wr, float ener0,
                                          const TwinNeigh &n, const HeapRef &heap, int &heap_size)
{
    const int nbx[4] = { 0, 0, 1, -1 }, nby[4] = { 1, -1, 0, 0 };
    const int t = threadIdx.x;
    const int ii = (i - wr < 0) ? 0 : i - wr, ij = (j - wr < 0) ? 0 : j - wr;
    const int pw = ((i + 1 + wr > nx) ? nx : i + 1 + wr) - ii;
    (void)ny;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int px = i + nbx[q], py = j + nby[q];
        const float new_ener = ener0 * lane_read(n.nsal, q);
        if (!__builtin_amdgcn_readlane(n.nfix, q) && new_ener < lane_read(n.nene, q)) {       // outside the image: nfix = 1 (wave-uniform test)
            if (t == 0) D.ene[py * nx + px] = ener0;
            const int wq = (py - ij) * pw + (px - ii);                            // the neighbour lies in the window
            if (heap_size < D.heap_cap) heap_push_wave(heap, heap_size, px, py, sm.a[wq], sm.b[wq], new_ener);
            else if (t == 0) D.ctl->overflow = 1;
        }
    }
    const int g = j * nx + i, wq = (j - ij) * pw + (i - ii);                      // :718-726
    if (t == 0 && lane_read(n.nene, 4) > ener0) {
        D.out[g] = sm.a[wq];
        D.out[pe + g] = sm.b[wq];
        D.ene[g] = ener0;
    }
}

__device__ __forceinline__ cand4 twin_lane_cand(const cand4 c, int lane)          // lane: wave-uniform
{
    cand4 r;
    r.x = lane_read(c.x, lane);
    r.y = lane_read(c.y, lane);
    r.z = lane_read(c.z, lane);
    r.w = lane_read(c.w, lane);
    return r;
}
__device__ __forceinline__ bool twin_same(const cand4 a, const cand4 b)
{
    return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
           __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
}

template <int K>
__global__ __launch_bounds__(K * kPatchThreads) void grow_kernel_twin(const GrowDir *__restrict__ dirs, int mode, int iteration, int wr,
                                                                     int max_iter_patch, int budget, int ntop)
{
    static_assert(K >= 2 && K <= kTwinMax, "slots per workgroup");
    __shared__ TwinShared<K> S;
    extern __shared__ __attribute__((aligned(16))) unsigned char grow_heap_top[];     // ntop queue elements (dynamic LDS, host-sized)
    const GrowDir D = dirs[blockIdx.x];
    const HeapRef heap{ (heap_ptr)D.heap, (heap_lds_ptr)(reinterpret_cast<cand4 *>(grow_heap_top)), ntop };
    heap_top_in(heap, D.ctl->heap_size);
    if (threadIdx.x < 8 + 4 * kTwinMax) S.ctl[threadIdx.x] = 0;
    __syncthreads();
    const int tid = threadIdx.x, slot = tid >> 7, t = tid & (kPatchThreads - 1), lane = tid & 63;
    GrowShared &sm = S.h[slot];
    const int nx = D.nx, ny = D.ny;
    const size_t pe = D.pe;
    int heap_size = 0, seed_pos = 0;
    long long pops = 0, solves = 0, fills = 0;
    long long cyc[3] = { 0, 0, 0 };
    if (tid < 64) {                                 // the first wavefront runs the queue; its lanes hold the same counters
        heap_size = D.ctl->heap_size;
        seed_pos = D.ctl->seed_pos;
    }
    int done = 0, windows = 0;
    GrowStamp gs{ false, 0 };
    const int wr_eff = mode == 0 ? 1 : wr;
    while (windows < budget) {
        // ---------------- the queue: pop A (as grow_body), then write down the expected next pops
        cand4 peek = { 0.0f, 0.0f, 0.0f, 0.0f };                                  // lane L < 7: queue element L (wave 0)
        unsigned seq_nodes = 0u, seq_slots = 0u;                                  // expected pops, 3 + 2 bits each: queue element; 0 = stale / s = slot s's window
        int nseq = 0, nguess = 0;
        if (tid < 64) {
            int have = 0, g = 0;
            const long long c0 = clock64();
            if (mode == 0) {
                if (seed_pos < D.nseeds) {
                    g = D.seeds[seed_pos++];
                    if (tid == 0) {
                        D.out[g] = D.seed_flow[g];
                        D.out[pe + g] = D.seed_flow[pe + g];
                        D.fixed[g] = 1;
                    }
                    have = 1;
                }
            } else {
                while (heap_size > 0) {
                    const cand4 c = heap.ld(0);
                    const unsigned cij = __float_as_uint(c.w);
                    const int ci = (int)(cij & 0xffffu), cj = (int)(cij >> 16);
                    g = cj * nx + ci;
                    const int was_fixed = D.fixed[g];                             // in flight during the sift-down
                    heap_remove_top_wave(heap, heap_size);
                    ++pops;
                    if (was_fixed) continue;
                    if (tid == 0) {
                        D.fixed[g] = 1;                                           // :926-932
                        D.out[g] = c.y;
                        D.out[pe + g] = c.z;
                        D.ene[g] = c.x;
                    }
                    have = 1;
                    break;
                }
                int room = budget - windows - 1;                                  // windows this launch may still process beside A
                if (room > K - 1) room = K - 1;
                if (have && heap_size > 0 && room > 0 && !D.shadow) {
                    // the three top levels of the queue and the state of their pixels, lane L <-> element L: one round trip
                    const bool pv = lane < kTwinPeek && lane < heap_size;
                    peek = heap.ld(pv ? lane : 0);
                    const unsigned pij = __float_as_uint(peek.w);
                    const int pgi = (int)(pij & 0xffffu), pgj = (int)(pij >> 16);
                    const int pg = pgj * nx + pgi;
                    const int pfx = (pg == g) ? 1 : D.fixed[pg];
                    // the frontier walk (wave-uniform): take the smallest energy of the frontier, add its children
                    int wi[kTwinMax] = { g % nx, 0, 0, 0 }, wj[kTwinMax] = { g / nx, 0, 0, 0 };   // pixels of this step's windows: A, guesses
                    int gnode[kTwinMax] = { 0, 0, 0, 0 };                         // the queue element each guess is
                    unsigned frontier = 1u;
                    const int npeek = heap_size < kTwinPeek ? heap_size : kTwinPeek;
#pragma unroll 1
                    for (int stepk = 0; stepk < kTwinSeq && frontier != 0u && nguess < room; ++stepk) {
                        int n = -1;
                        float best = 0.0f;
#pragma unroll
                        for (int q = 0; q < kTwinPeek; ++q) {
                            const float sq = lane_read(peek.x, q);
                            if (((frontier >> q) & 1u) && (n < 0 || sq < best)) { n = q; best = sq; }
                        }
                        frontier &= ~(1u << n);
                        const int ni = __builtin_amdgcn_readlane(pgi, n), nj = __builtin_amdgcn_readlane(pgj, n);
                        bool stale = __builtin_amdgcn_readlane(pfx, n) != 0;
                        bool apart = true;
#pragma unroll
                        for (int sl = 0; sl < kTwinMax; ++sl)
                            if (sl <= nguess) {
                                const int di = ni - wi[sl], dj = nj - wj[sl];
                                if (sl > 0 && di == 0 && dj == 0) stale = true;                // an earlier guess of this step fixes it
                                apart = apart && (di > 2 * wr || di < -2 * wr || dj > 2 * wr || dj < -2 * wr);
                            }
                        if (!stale) {
                            if (!apart) break;                                     // it must see the earlier windows' results: the list ends
                            ++nguess;
#pragma unroll
                            for (int sl = 1; sl < kTwinMax; ++sl)
                                if (sl == nguess) { wi[sl] = ni; wj[sl] = nj; gnode[sl] = n; }
                        }
                        seq_nodes |= (unsigned)n << (3 * nseq);
                        seq_slots |= (unsigned)(stale ? 0 : nguess) << (2 * nseq);
                        ++nseq;
                        if (2 * n + 1 < npeek) frontier |= 1u << (2 * n + 1);      // its children join the frontier (those that were loaded)
                        if (2 * n + 2 < npeek) frontier |= 1u << (2 * n + 2);
                        if (2 * n + 1 >= kTwinPeek && 2 * n + 1 < heap_size) break;   // children that were not loaded: what follows is unknown
                    }
                    while (nseq > 0 && ((seq_slots >> (2 * (nseq - 1))) & 3u) == 0u) --nseq;   // (stale pops after the last guess are the next step's)
                    // the guesses' candidates go to LDS: their pixels' fix (:926-932) is applied to the PLANES only when a guess is
                    // confirmed; until then the guess's own window takes it from here (harmonic_fill_window: fix_xy)
#pragma unroll
                    for (int s = 1; s < kTwinMax; ++s) {
                        if (s > nguess) continue;
                        const cand4 c = twin_lane_cand(peek, gnode[s]);
                        if (tid == 0) {
                            S.ctl[8 + 4 * s] = wi[s];
                            S.ctl[9 + 4 * s] = wj[s];
                            S.cand[s][0] = c.x;
                            S.cand[s][1] = c.y;
                            S.cand[s][2] = c.z;
                        }
                    }
                }
            }
            if (tid == 0) {
                cyc[0] += clock64() - c0;
                S.ctl[0] = have;
                S.ctl[1] = nguess;
                S.ctl[8] = g % nx;
                S.ctl[9] = g / nx;
            }
        }
        __syncthreads();
        if (!S.ctl[0]) { done = 1; break; }
        const int ng = S.ctl[1];                                                  // guesses of this step (uniform)
        const int i = S.ctl[8 + 4 * slot], j = S.ctl[9 + 4 * slot];
        const bool on = slot <= ng;

        // ---------------- the windows (add_neighbors :679-727 up to the candidates), one per slot
        const int ii = (i - wr_eff < 0) ? 0 : i - wr_eff, ij = (j - wr_eff < 0) ? 0 : j - wr_eff;          // get_index_patch
        const int ei = (i + 1 + wr_eff > nx) ? nx : i + 1 + wr_eff, ej = (j + 1 + wr_eff > ny) ? ny : j + 1 + wr_eff;
        const int pw = on ? ei - ii : 1, ph = on ? ej - ij : 1;
        const bool act = on && t < pw * ph;
        const int p = (ij + (act ? t / pw : 0)) * nx + ii + (act ? t % pw : 0);
        bool fill = iteration == 0;
        if (iteration != 0) {                                                     // check_trustable_patch :639-656, per slot:
            // every wavefront votes on its part of its slot's window, one barrier, the slot's two votes are read back (a
            // __syncthreads_or per slot is two barriers and an LDS atomic each)
            const bool mine = act && D.trust[p] == 0;
            const unsigned long long vote = __builtin_amdgcn_ballot_w64(mine);
            if (lane == 0) S.untrusted[tid >> 6] = vote != 0ull ? 1 : 0;
            __syncthreads();
            fill = (S.untrusted[2 * slot] | S.untrusted[2 * slot + 1]) != 0;
        }
        TwinNeigh nb[K];
        if (tid < 64) {                                                           // every window's neighbour state, in flight during the windows
#pragma unroll
            for (int s = 0; s < K; ++s)
                if (s <= ng) {
                    twin_neigh_load(D, nx, ny, S.ctl[8 + 4 * s], S.ctl[9 + 4 * s], nb[s]);
                    if (s > 0 && lane == 4) nb[s].nene = S.cand[s][0];            // the centre's energy once its fix is in (:932)
                }
        }
        float u1 = 0.0f, u2 = 0.0f;
        const long long c0 = clock64();
        harmonic_fill_window(sm, D.out, pe, D.fixed, nx, ii, ij, pw, ph, u1, u2, gs, t, on && fill,
                             (slot != 0 && on) ? j * nx + i : -1, S.cand[slot][1], S.cand[slot][2]);
        if (!fill && act) {
            u1 = D.planes[L_U1 * pe + p];
            u2 = D.planes[L_U2 * pe + p];
        }
        const long long c1 = clock64();
        const int iiw = (i - wr_eff < 0) ? -(i - wr_eff) : 0, ijw = (j - wr_eff < 0) ? -(j - wr_eff) : 0;    // get_relative_index_weight
        TwinArgs tw{ t, on, slot != 0, slot, (2u << ng) - 1u, S.h, 0.0f, 0.0f };
        const float ener = window_solve<true>(sm, D.planes, pe, nx, ny, ii, ij, ei, ej, max_iter_patch, u1, u2, D.weight, iiw, ijw, wr_eff,
                                              slot == 0 ? D.shadow : 0, gs, &tw);
        const long long c2 = clock64();
        if (t == 0) {
            S.ctl[11 + 4 * slot] = __float_as_int(ener);
            S.filled[slot] = (on && fill) ? 1 : 0;
        }

        // ---------------- A's candidates; then the expected pops are replayed on the real queue
        if (tid < 64) {
            const float enerA = lane_read(ener, 0);                               // thread 0 summed A's energy
            twin_push(D, S.h[0], pe, nx, ny, S.ctl[8], S.ctl[9], wr_eff, enerA, nb[0], heap, heap_size);
        }
        int alive = 1;                                                            // wave 0: the replay has not met a mismatch yet
        int q_next = 0;                                                           // wave 0: next entry of the expected sequence
#pragma unroll 1
        for (int s = 1; s <= ng; ++s) {                                           // uniform: ng comes from LDS
            if (tid < 64) {
                int commit = 0;
                while (alive && q_next < nseq) {
                    const int node = (int)((seq_nodes >> (3 * q_next)) & 7u), sl = (int)((seq_slots >> (2 * q_next)) & 3u);
                    if (sl != 0 && sl != s) break;                                // (cannot happen: guesses appear in slot order)
                    const cand4 want = twin_lane_cand(peek, node);
                    const bool same = heap_size > 0 && twin_same(heap.ld(0), want);
                    if (!same) { alive = 0; break; }
                    heap_remove_top_wave(heap, heap_size);                        // the stale candidate goes, or the guess is popped
                    ++pops;
                    ++q_next;
                    if (sl == s) { commit = 1; break; }
                }
                if (!commit) alive = 0;                                           // (a guess that is not the next pop leaves nothing behind)
                if (commit && tid == 0) {                                         // its pixel's fix, :926-932
                    const int gs_ = S.ctl[9 + 4 * s] * nx + S.ctl[8 + 4 * s];
                    D.fixed[gs_] = 1;
                    D.out[gs_] = S.cand[s][1];
                    D.out[pe + gs_] = S.cand[s][2];
                    D.ene[gs_] = S.cand[s][0];
                }
                if (tid == 0) S.ctl[10 + 4 * s] = commit;
            }
            __syncthreads();
            const int commit = S.ctl[10 + 4 * s];
            if (commit) {                                                         // uniform
                if (slot == s && act) {                                           // the guess's window, held back until now
                    D.planes[L_U1 * pe + p] = u1;
                    D.planes[L_U2 * pe + p] = u2;
                    D.planes[L_DIV1 * pe + p] = tw.d1;
                    D.planes[L_DIV2 * pe + p] = tw.d2;
                }
                if (tid < 64) {
                    const float enerS = __int_as_float(S.ctl[11 + 4 * s]);
                    TwinNeigh n = nb[1];
#pragma unroll
                    for (int q = 2; q < K; ++q) if (q == s) n = nb[q];
                    twin_push(D, S.h[s], pe, nx, ny, S.ctl[8 + 4 * s], S.ctl[9 + 4 * s], wr_eff, enerS, n, heap, heap_size);
                }
                if (tid == 0) { ++solves; fills += S.filled[s]; }
                ++windows;
            }
        }
        if (tid == 0) {
            ++solves;
            fills += S.filled[0];
            cyc[0] += clock64() - c2;
            cyc[1] += c1 - c0;
            cyc[2] += c2 - c1;
            if (mode == 0) {                                                      // :778-780
                const int g = S.ctl[9] * nx + S.ctl[8];
                D.out[g] = __builtin_nanf("");
                D.out[pe + g] = __builtin_nanf("");
                D.fixed[g] = 0;
            }
        }
        ++windows;
        __syncthreads();                                                          // S.ctl is rewritten by the next step's queue phase
    }
    if (tid == 0) S.ctl[2] = heap_size;
    __syncthreads();                                // (every queue operation of the first wavefront is behind its fence)
    heap_top_out(heap, S.ctl[2]);
    if (tid == 0) {
        if (!done) done = (mode == 0) ? (seed_pos >= D.nseeds) : (heap_size == 0);
        D.ctl->heap_size = heap_size;
        D.ctl->seed_pos = seed_pos;
        D.ctl->done = done;
        D.ctl->pops += pops;
        D.ctl->solves += solves;
        D.ctl->fills += fills;
        D.ctl->cyc_queue += cyc[0];
        D.ctl->cyc_fill += cyc[1];
        D.ctl->cyc_solve += cyc[2];
    }
}

}  // namespace faldoi
