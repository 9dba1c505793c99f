// grow_twin_kernel.hpp -- one growing, two windows at a time where the pop order allows it (round 4).
//
// A growing is sequential: pop the best candidate, fix its pixel, fill and minimise the window around it, push its neighbours, pop
// again (src/local_faldoi.cpp:891-937, :679-727).  One workgroup runs one growing, and a lone wavefront issues one instruction per
// ~5 cycles: a window costs ~53 k cycles of which ~41 k are the fill and the minimisation (profiles/r04_growing_detail.md).
//
// What CAN run side by side without changing a bit: after candidate A has been popped, look at the queue's new top B.  If B's pixel is
// not fixed and B's window does not overlap A's (|di| > 2 wr or |dj| > 2 wr), then -- provided B is still the top once A's candidates
// have been pushed -- the sequential algorithm pops B next, and B's window reads nothing A's window wrote.  So a 256-thread workgroup
// computes both windows at once, half 0 (threads 0..127) A's and half 1 B's, with B's results held back; wave 0 pushes A's
// candidates exactly as before and then looks at the top: if it is B (the same element, bit for bit), B is popped, its results are
// stored and its candidates pushed -- the heap has seen exactly the sequential operations, in the sequential order --; if not, B's
// speculative fix (fixed / out / ene of its pixel) is undone and nothing of it remains.  On the example pairs 21-31 % of the windows
// are served this way (oracle study hook orc_twin_stats: the top is stale in 20-28 % of the steps, overlaps in 19-30 %, is displaced
// by a push in 0.5 %).
//
// Both halves execute the same workgroup barriers: the window functions take the half's thread index and an `on` flag
// (harmonic_fill_window, window_solve<true> in grow_kernels.hpp), and the iteration loop of window_solve runs while either half goes on.
// TV-l2 coupled functionals (-m 0 / 1), no image partitions (their out-of-window divergence statements would break the disjointness).
#pragma once
#include "grow_kernels.hpp"

#pragma clang fp contract(off)

namespace faldoi {

struct TwinShared {
    GrowShared h[2];
    int ctl[16];          // 0 have A | 1, 2 A's pixel | 3 twin | 4, 5 B's pixel | 6 commit B | 8, 9 window energies (float bits) | 10, 11 filled
};

// The four neighbours' state of a pixel, read by wave 0 ahead of the window computation (insert_candidates :497-537 needs it)
struct TwinNeigh {
    int nfix[4];
    float nene[4], nsal[4], cene;
};
__device__ __forceinline__ void twin_neigh_load(const GrowDir &D, int nx, int ny, int i, int j, TwinNeigh &n)
{
    const int nbx[4] = { 0, 0, 1, -1 }, nby[4] = { 1, -1, 0, 0 };                 // insert_candidates :501-505
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int px = i + nbx[q], py = j + nby[q];
        const bool inside = px >= 0 && px < nx && py >= 0 && py < ny;
        const int g = inside ? py * nx + px : j * nx + i;
        const int f = D.fixed[g];
        n.nfix[q] = inside ? f : 1;
        n.nene[q] = D.ene[g];
        n.nsal[q] = D.sal ? D.sal[g] : 1.0f;
    }
    n.cene = D.ene[j * nx + i];
}
// insert_candidates (:497-537) + the centre's update (:718-726) for one window, by wave 0; sm = the LDS block of the half that
// minimised it (flow of the window in sm.a / sm.b), ener0 = its energy (wave-uniform)
__device__ __forceinline__ void twin_push(const GrowDir &D, GrowShared &sm, size_t pe, int nx, int i, int j, int ii, int ij, int pw,
                                          float ener0, const TwinNeigh &n, const HeapRef &heap, int &heap_size)
{
    const int nbx[4] = { 0, 0, 1, -1 }, nby[4] = { 1, -1, 0, 0 };
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int px = i + nbx[q], py = j + nby[q];
        const float new_ener = ener0 * n.nsal[q];
        if (!n.nfix[q] && new_ener < n.nene[q]) {                                 // outside the image: nfix = 1 (wave-uniform test)
            if (t == 0) D.ene[py * nx + px] = ener0;
            const int wq = (py - ij) * pw + (px - ii);                            // the neighbour lies in the window
            if (heap_size < D.heap_cap) heap_push_wave(heap, heap_size, px, py, sm.a[wq], sm.b[wq], new_ener);
            else if (t == 0) D.ctl->overflow = 1;
        }
    }
    const int g = j * nx + i, wq = (j - ij) * pw + (i - ii);                      // :718-726
    if (t == 0 && n.cene > ener0) {
        D.out[g] = sm.a[wq];
        D.out[pe + g] = sm.b[wq];
        D.ene[g] = ener0;
    }
}

__global__ __launch_bounds__(2 * kPatchThreads) void grow_kernel_twin(const GrowDir *__restrict__ dirs, int mode, int iteration, int wr,
                                                                     int max_iter_patch, int budget, int ntop)
{
    __shared__ TwinShared S;
    extern __shared__ __attribute__((aligned(16))) unsigned char grow_heap_top[];     // ntop queue elements (dynamic LDS, host-sized)
    const GrowDir D = dirs[blockIdx.x];
    const HeapRef heap{ (heap_ptr)D.heap, (heap_lds_ptr)(reinterpret_cast<cand4 *>(grow_heap_top)), ntop };
    heap_top_in(heap, D.ctl->heap_size);
    __syncthreads();
    const int tid = threadIdx.x, half = tid >> 7, t = tid & (kPatchThreads - 1);
    GrowShared &sm = S.h[half];
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
        // ---------------- the queue: pop A (as grow_body), look at the new top B
        cand4 cB = { 0.0f, 0.0f, 0.0f, 0.0f }, cS = { 0.0f, 0.0f, 0.0f, 0.0f };
        int stale_first = 0;                                                      // the top is a stale candidate cS; B is its guessed successor
        float old_u = 0.0f, old_v = 0.0f, old_e = 0.0f;                           // B's pixel before the speculative fix (thread 0)
        if (tid < 64) {
            int have = 0, g = 0, twin = 0, gB = 0;
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
                if (have && heap_size > 0 && windows + 2 <= budget && !D.shadow) {
                    cB = heap.ld(0);
                    unsigned bij = __float_as_uint(cB.w);
                    int bi = (int)(bij & 0xffffu), bj = (int)(bij >> 16);
                    gB = bj * nx + bi;
                    int fB = (gB == g) ? 1 : D.fixed[gB];
                    if (fB && heap_size > 1) {
                        // The top is stale (its pixel is fixed: the sequential loop pops it and goes on, :912-916).  Its successor is one
                        // of the root's children -- a guess, checked like everything else once A's candidates are in: first the top
                        // must still be this stale element (it is popped: it is stale whatever happens), then the new top must be B.
                        cS = cB;
                        stale_first = 1;
                        const cand4 l = heap.ld(1), r = heap.ld(heap_size > 2 ? 2 : 1);
                        cB = r.x < l.x ? r : l;
                        bij = __float_as_uint(cB.w);
                        bi = (int)(bij & 0xffffu);
                        bj = (int)(bij >> 16);
                        gB = bj * nx + bi;
                        fB = (gB == g) ? 1 : D.fixed[gB];
                    }
                    const int di = bi - g % nx, dj = bj - g / nx;
                    const bool apart = di > 2 * wr || di < -2 * wr || dj > 2 * wr || dj < -2 * wr;
                    if (apart && fB == 0) {
                        twin = 1;
                        if (tid == 0) {                                           // B's fix (:926-932), undone below if B is not the next pop
                            old_u = D.out[gB];
                            old_v = D.out[pe + gB];
                            old_e = D.ene[gB];
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the old values are in before the new ones leave)
                            D.fixed[gB] = 1;
                            D.out[gB] = cB.y;
                            D.out[pe + gB] = cB.z;
                            D.ene[gB] = cB.x;
                        }
                    }
                }
            }
            if (tid == 0) {
                cyc[0] += clock64() - c0;
                S.ctl[0] = have;
                S.ctl[1] = g % nx;
                S.ctl[2] = g / nx;
                S.ctl[3] = twin;
                S.ctl[4] = gB % nx;
                S.ctl[5] = gB / nx;
            }
        }
        __syncthreads();
        if (!S.ctl[0]) { done = 1; break; }
        const int twin = S.ctl[3];
        const int iA = S.ctl[1], jA = S.ctl[2], iB = S.ctl[4], jB = S.ctl[5];
        const int i = half ? iB : iA, j = half ? jB : jA;
        const bool on = half == 0 || twin != 0;

        // ---------------- the two windows (add_neighbors :679-727 up to the candidates)
        const int ii = (i - wr_eff < 0) ? 0 : i - wr_eff, ij = (j - wr_eff < 0) ? 0 : j - wr_eff;          // get_index_patch
        const int ei = (i + 1 + wr_eff > nx) ? nx : i + 1 + wr_eff, ej = (j + 1 + wr_eff > ny) ? ny : j + 1 + wr_eff;
        const int pw = ei - ii, ph = ej - ij;
        const bool act = on && t < pw * ph;
        const int p = (ij + (act ? t / pw : 0)) * nx + ii + (act ? t % pw : 0);
        bool fill = iteration == 0;
        if (iteration != 0) {                                                     // check_trustable_patch :639-656, per half
            const int fa = __syncthreads_or(half == 0 && act && D.trust[p] == 0);
            const int fb = __syncthreads_or(half == 1 && act && D.trust[p] == 0);
            fill = (half ? fb : fa) != 0;
        }
        TwinNeigh nA, nB;
        if (tid < 64) {                                                           // both windows' neighbour state, in flight during the windows
            twin_neigh_load(D, nx, ny, iA, jA, nA);
            if (twin) twin_neigh_load(D, nx, ny, iB, jB, nB);
        }
        float u1 = 0.0f, u2 = 0.0f;
        const long long c0 = clock64();
        harmonic_fill_window(sm, D.out, pe, D.fixed, nx, ii, ij, pw, ph, u1, u2, gs, t, on && fill);
        if (!fill && act) {
            u1 = D.planes[L_U1 * pe + p];
            u2 = D.planes[L_U2 * pe + p];
        }
        const long long c1 = clock64();
        const int iiw = (i - wr_eff < 0) ? -(i - wr_eff) : 0, ijw = (j - wr_eff < 0) ? -(j - wr_eff) : 0;    // get_relative_index_weight
        TwinArgs tw{ t, on, half ? true : (twin != 0), half == 1, &S.h[1 - half], 0.0f, 0.0f };
        const float ener = window_solve<true>(sm, D.planes, pe, nx, ny, ii, ij, ei, ej, max_iter_patch, u1, u2, D.weight, iiw, ijw, wr_eff,
                                              half == 0 ? D.shadow : 0, gs, &tw);
        const long long c2 = clock64();
        if (t == 0) {
            S.ctl[8 + half] = __float_as_int(ener);
            S.ctl[10 + half] = fill ? 1 : 0;
        }

        // ---------------- A's candidates, then: is B the next pop?
        if (tid < 64) {
            const float enerA = lane_read(ener, 0);                               // thread 0 summed A's energy
            twin_push(D, S.h[0], pe, nx, iA, jA, (iA - wr_eff < 0) ? 0 : iA - wr_eff, (jA - wr_eff < 0) ? 0 : jA - wr_eff,
                      ((iA + 1 + wr_eff > nx) ? nx : iA + 1 + wr_eff) - ((iA - wr_eff < 0) ? 0 : iA - wr_eff), enerA, nA, heap, heap_size);
            int commit = 0;
            if (twin) {
                auto is_top = [&](const cand4 c) {
                    if (heap_size <= 0) return false;
                    const cand4 top = heap.ld(0);
                    return __float_as_uint(top.x) == __float_as_uint(c.x) && __float_as_uint(top.y) == __float_as_uint(c.y) &&
                           __float_as_uint(top.z) == __float_as_uint(c.z) && __float_as_uint(top.w) == __float_as_uint(c.w);
                };
                bool same = true;
                if (stale_first) {
                    same = is_top(cS);
                    if (same) {                                                   // the stale candidate goes, as in the sequential loop
                        heap_remove_top_wave(heap, heap_size);
                        ++pops;
                    }
                }
                same = same && is_top(cB);
                if (same) {
                    heap_remove_top_wave(heap, heap_size);
                    ++pops;
                    commit = 1;
                } else if (tid == 0) {                                            // B is not the next pop: its pixel as it was
                    const int gB = jB * nx + iB;
                    D.fixed[gB] = 0;
                    D.out[gB] = old_u;
                    D.out[pe + gB] = old_v;
                    D.ene[gB] = old_e;
                }
            }
            if (tid == 0) S.ctl[6] = commit;
        }
        __syncthreads();
        const int commit = twin ? S.ctl[6] : 0;
        if (commit) {                                                             // uniform
            if (half == 1 && act) {                                               // B's window, held back until now
                D.planes[L_U1 * pe + p] = u1;
                D.planes[L_U2 * pe + p] = u2;
                D.planes[L_DIV1 * pe + p] = tw.d1;
                D.planes[L_DIV2 * pe + p] = tw.d2;
            }
            if (tid < 64) {
                const float enerB = __int_as_float(S.ctl[9]);
                twin_push(D, S.h[1], pe, nx, iB, jB, (iB - wr_eff < 0) ? 0 : iB - wr_eff, (jB - wr_eff < 0) ? 0 : jB - wr_eff,
                          ((iB + 1 + wr_eff > nx) ? nx : iB + 1 + wr_eff) - ((iB - wr_eff < 0) ? 0 : iB - wr_eff), enerB, nB, heap, heap_size);
            }
        }
        if (tid == 0) {
            ++solves;
            fills += S.ctl[10];
            if (commit) { ++solves; fills += S.ctl[11]; }
            cyc[0] += clock64() - c2;
            cyc[1] += c1 - c0;
            cyc[2] += c2 - c1;
            if (mode == 0) {                                                      // :778-780
                const int g = jA * nx + iA;
                D.out[g] = __builtin_nanf("");
                D.out[pe + g] = __builtin_nanf("");
                D.fixed[g] = 0;
            }
        }
        windows += commit ? 2 : 1;
        __syncthreads();                                                          // S.ctl is rewritten by the next step's queue phase
    }
    if (tid == 0) S.ctl[7] = heap_size;
    __syncthreads();                                // (every queue operation of the first wavefront is behind its fence)
    heap_top_out(heap, S.ctl[7]);
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
