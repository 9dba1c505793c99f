// grow_kernels.hpp -- the local step of FALDOI (match growing) resident on the device.
//
// Reference: src/local_faldoi.cpp: insert_initial_seeds :748-796, local_growing :891-1039,
// add_neighbors :679-727, insert_candidates :497-537, fb_consistency_check :167-190,
// delete_not_trustable_candidates :283-311, prepare_data_for_growing :860-870 (TV-l2 coupled
// functional, no partitions, no occlusions).
//
// The growing is sequential by definition: the candidate with the smallest energy is popped, its
// pixel is fixed, an 11x11 window around it is (re)filled and minimised, and its four neighbours
// are pushed with the new energy; every step reads what the previous ones wrote.  To return the
// reference's flow bit for bit the order is kept: ONE workgroup (two wavefronts) owns one growing
// (one direction of one image pair) from the first pop to the last, with the queue, the flow and
// the energies in HBM and the window in registers / LDS.  There is no host round trip per
// candidate.  Parallelism comes from independent growings: forward and backward of a pair run as
// two workgroups of the same launch, and a batch of pairs fills the 256 CUs.
//
// A launch processes at most `budget` windows per growing and returns (every wave reaches the
// exit), the host relaunches until the queue is empty.
#pragma once
#include "faldoi_device.hpp"
#include "grow_heap.hpp"
#include "local_kernels.hpp"

#pragma clang fp contract(off)

namespace faldoi {

struct GrowCtl {                      // per growing, in HBM
    int heap_size;
    int seed_pos;                     // next entry of the seed list (seed phase)
    int done;                         // 1 when the phase has nothing left to process
    int overflow;                     // set to 1 if a push did not fit (cannot happen: heap_cap is the worst case)
    long long pops, solves, fills;    // statistics
    long long cyc_queue, cyc_fill, cyc_solve;   // shader-clock cycles spent by thread 0 in each part
};

struct GrowDir {                      // per growing: device pointers (array in HBM, one entry per workgroup)
    float *planes;                    // [L_COUNT][pe]: source frame, target frame + gradient, flow, duals, divergences
    int *fixed;                       // OpticalFlowData::fixed_points
    int *trust;                       // OpticalFlowData::trust_points
    float *ene;                       // ene_Go / ene_Ba
    float *out;                       // oft0 / oft1: [2][pe]
    const float *sal;                 // saliency or nullptr (= 1)
    const float *weight;              // gaussian1Dweight(w_radio) for the weighted functional (-m 1), else nullptr
    Candidate *heap;
    GrowCtl *ctl;
    const int *seeds;                 // pixel indices of the finite seeds, row-major order
    const float *seed_flow;           // [2][pe]
    int nseeds;
    int heap_cap;
    int nx, ny;                       // size of the domain this growing runs on (the frame, or one partition of it)
    int shadow;                       // 1: keep the duals and the out-of-window divergence statements in HBM (partitions)
    int method;                       // -m: 0 / 1 TV-l2 coupled (plain / weighted), 2..7 the functionals of grow_nl_kernels.hpp
    size_t pe;                        // plane stride of `planes`, `out`, `seed_flow`
    float *duals;                     // -m 2, 3, 6, 7: the non-local dual values p | q, [48][pe] (persistent across windows)
    const float *wp;                  // ... and the constant non-local weights, [24][pe]
    int detail;                       // measurement aid (FALDOI_GROW_DETAIL=1): thread 0 of growing 0 splits its cycles by sub-phase
};

// FALDOI_GROW_DETAIL=1 (tests/tools/time_growing.py --detail): where the committing wavefront's cycles go, finer than GrowCtl's
// three sums.  Thread 0 of the launch's first growing adds the shader-clock cycles since its previous mark to a device word
// (fire-and-forget atomics).  Slots: 0 pop | 1 trust test + neighbour prefetch | 2 fill: window loaded | 3 fill: pyramid built |
// 4 fill: level initialisation and hole masks | 5 fill: sweeps of level 0 | 6 fill: sweeps of the coarser levels | 7 fill: result
// handed over | 8 solve: bicubic warp + set-up | 9 solve: iterations | 10 solve: flow stored + energy | 11 candidates pushed
constexpr int kGrowDetailSlots = 12;
__device__ unsigned long long g_grow_detail[kGrowDetailSlots];
struct GrowStamp {
    bool on;
    long long last;
    __device__ __forceinline__ void start() { if (on) last = clock64(); }
    __device__ __forceinline__ void mark(int slot)
    {
        if (on) {
            const long long now = clock64();
            atomicAdd(&g_grow_detail[slot], (unsigned long long)(now - last));
            last = now;
        }
    }
};

struct GrowShared {
    float in[2][kElapPyr];            // harmonic fill: data pyramid (NaN = hole), per channel
    float fin[2][kElapPyr];           // harmonic fill: solution pyramid
    float ver[2][4][kElapPyr];        // harmonic fill: the level before / after each of the three sweeps
    float a[kPatchThreads], b[kPatchThreads];     // window exchange of ubar / u
    float x[4][kPatchThreads];        // duals of the current iteration
    alignas(16) float t[kPatchThreads];   // energy terms
    int err[4];                       // max |du|^2 (float bits) per wavefront, double-buffered
    int res[2];                       // harmonic fill: offset of each channel's level-0 result from ver[c][0]
    int cur[4];                       // {have, i, j, pixel}
    int vote[2];                      // trust test: per wavefront, its part of the window has a pixel that is not trusted
};

}  // namespace faldoi
#include "grow_nl_kernels.hpp"
namespace faldoi {

// ---------------------------------------------------------------------------------------------
// interpolate_poisson on the window.  Wavefront c fills flow channel c on its own (no workgroup
// barrier inside: LDS operations of one wavefront execute in order).
//
// The sweeps of src/elap_recsep.c:50-69 update the holes in row-major order and every update sees
// its left and upper neighbours already updated, its right and lower ones not yet.  Hence (a) the
// pixels of one anti-diagonal are independent, and (b) sweep k+1 may work on diagonal d as soon as
// sweep k has finished diagonal d+1.  Lane group g (16 lanes, lane i = column i) runs sweep g+1 on
// diagonal step-3g: the three sweeps of a level are in flight together, each writing its own version
// of the level (ver[g+1]; ver[0] is the initial state), so the values, their order of dependence and
// the early exit (`maxupdate < 1e-10` after a sweep: the later versions are simply not used) are
// the reference's.  Returns this thread's filled (u1, u2) (threads outside the window: 0).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_read(float v, int lane)    // lane: wave-uniform
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// max of a wavefront's values, all >= 0 (a shifted-in 0 is neutral; `a > b ? a : b` on NaN-free operands is order-independent): four
// row shifts leave each 16-lane row's maximum in its last lane, two row broadcasts carry them to lane 63
__device__ __forceinline__ float wave_max_nonneg(float v)
{
    float o;
    o = __builtin_amdgcn_update_dpp(0.0f, v, 0x111, 0xf, 0xf, true); v = o > v ? o : v;      // row_shr:1
    o = __builtin_amdgcn_update_dpp(0.0f, v, 0x112, 0xf, 0xf, true); v = o > v ? o : v;      // row_shr:2
    o = __builtin_amdgcn_update_dpp(0.0f, v, 0x114, 0xf, 0xf, true); v = o > v ? o : v;      // row_shr:4
    o = __builtin_amdgcn_update_dpp(0.0f, v, 0x118, 0xf, 0xf, true); v = o > v ? o : v;      // row_shr:8
    o = __builtin_amdgcn_update_dpp(0.0f, v, 0x142, 0xa, 0xf, false); v = o > v ? o : v;     // row_bcast:15 into rows 1 and 3
    o = __builtin_amdgcn_update_dpp(0.0f, v, 0x143, 0xc, 0xf, false); v = o > v ? o : v;     // row_bcast:31 into rows 2 and 3
    return lane_read(v, 63);
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// t: the thread's index within the 128 threads that work on this window (threadIdx.x, or threadIdx.x & 127 in the twin kernel);
// on = false (twin kernel: this half has no window to fill in this step): nothing but the workgroup barrier at the end.
__device__ __forceinline__ void harmonic_fill_window(GrowShared &sm, const float *out, size_t pe, const int *fixed, int nx,
                                                     int ii, int ij, int pw, int ph, float &u1, float &u2, GrowStamp &gs,
                                                     int t, bool on = true, int fix_xy = -1, float fix_u = 0.0f, float fix_v = 0.0f)
{
    // fix_xy >= 0 (twin kernel, a guessed window): that pixel counts as fixed with the flow (fix_u, fix_v) -- the candidate's own pixel,
    // whose fix (:926-932) is only written to the planes when the guess is confirmed
    const int c = t >> 6, lane = t & 63;
    const int g = lane >> 4, i = lane & 15;
    if (on) {
    int ws[kElapLevels], hs[kElapLevels], off[kElapLevels];
    ws[0] = pw; hs[0] = ph; off[0] = 0;
#pragma unroll
    for (int s = 1; s < kElapLevels; ++s) {
        ws[s] = (ws[s - 1] + 1) / 2;                               // ceil(w / 2.0), src/elap_recsep.c:195-196
        hs[s] = (hs[s - 1] + 1) / 2;
        off[s] = off[s - 1] + ws[s - 1] * hs[s - 1];
    }
    float *in = sm.in[c], *fin = sm.fin[c];
    const float *src = out + (size_t)c * pe;
    for (int q = lane; q < pw * ph; q += 64) {                      // copy_fixed_coordinates + :341-355
        const int xy = (ij + q / pw) * nx + ii + q % pw;
        const int fx = fixed[xy];                                   // both loads in flight, then the selection
        const float val = src[xy];
        in[q] = (xy == fix_xy) ? (c ? fix_v : fix_u) : ((fx == 1) ? val : __builtin_nanf(""));
    }
    wave_sync();
    gs.mark(2);
#pragma unroll
    for (int s = 1; s < kElapLevels; ++s) {                         // zoom_out_by_factor_two :160-175
        const int iw = ws[s - 1], ih = hs[s - 1], ow = ws[s];
        if (iw * ih == 1) {                                         // 1x1 from 1x1: the four clamped samples are the pixel itself
            if (lane == 0) {
                const float v = in[off[s - 1]];
                float m = 0;
                m += v; m += v; m += v; m += v;
                in[off[s]] = __builtin_isfinite(v) ? m / 4 : __builtin_nanf("");
            }
            wave_sync();
            continue;
        }
        if (lane < ow * hs[s]) {
            const int zi = lane % ow, zj = lane / ow;
            const float *x = in + off[s - 1];
            const float a4[4] = { px_clamp(x, iw, ih, 2 * zi, 2 * zj), px_clamp(x, iw, ih, 2 * zi + 1, 2 * zj),
                                  px_clamp(x, iw, ih, 2 * zi, 2 * zj + 1), px_clamp(x, iw, ih, 2 * zi + 1, 2 * zj + 1) };
            float m = 0;
            int cx = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (__builtin_isfinite(a4[q])) { m += a4[q]; cx += 1; }
            in[off[s] + lane] = cx ? m / cx : __builtin_nanf("");
        }
        wave_sync();
    }
    gs.mark(3);
    const float *coarse = fin;                                      // solution of the level below (level-relative indexing)
    // Levels 2.. of a window of at most 11x11 have at most 3x3 pixels.  The pipelined sweeps below cost by their steps
    // (w + h - 1 + 6 per level, ~290 cycles each: 11 steps for 3x3, 9 for 2x2), not by their pixels; for so few pixels the three
    // sweeps run serially instead, in registers, every lane computing the same thing: the level sits in a 3x3 grid (lane e < 9 loads
    // pixel (e % 3, e / 3)), nine `readlane`s spread it, an update is the statement of src/elap_recsep.c:50-69 in its written
    // order with getpixel_1's clamping resolved per position (left / up at compile time, right / down by the level's size), the
    // early exit is the reference's (`maxupdate < 1e-10` after a sweep).  One rolled loop over these levels: its body exists once.
    {
        int wc = 1, hc = 1;                                         // size of the level below
#pragma unroll 1
        for (int s = kElapLevels - 1; s >= 2; --s) {
            const int w = (pw + (1 << s) - 1) >> s, h = (ph + (1 << s) - 1) >> s;      // ceil(ceil(w / 2) / 2 ...) = ceil(w / 2^s)
            const int o = s == 2 ? off[2] : s == 3 ? off[3] : s == 4 ? off[4] : s == 5 ? off[5] : off[kElapLevels - 1];
            static_assert(kElapLevels == 7 && kPatchMax == 11, "levels 2..6 of an at most 11x11 window");
            const float *x = in + o;
            if (w * h == 1) {
                // a single pixel: its four neighbours clamp to itself; the three sweeps run in registers
                if (lane == 0) {
                    float v = x[0];
                    if (!__builtin_isfinite(v)) {
                        v = (s < kElapLevels - 1) ? coarse[0] : 0.0f;
                        for (int it = 0; it < 3; ++it) {
                            const float lap = -4 * v + v + v + v + v;
                            const float nw = v + 0.4f * lap;
                            const float update = __builtin_fabsf(v - nw);
                            v = nw;
                            if ((double)update < 1e-10) break;
                        }
                    }
                    fin[o] = v;
                }
                coarse = fin + o;
                wc = 1;
                hc = 1;
                wave_sync();
                continue;
            }
            const int j3 = (lane * 11) >> 5, i3 = lane - 3 * j3;   // lane e < 9: (e % 3, e / 3)
            const bool valid = lane < 9 && i3 < w && j3 < h;
            const int q = valid ? j3 * w + i3 : 0;
            const float xv = x[q];
            float v = xv;
            float coarse_v = 0.0f;                                  // coarsest level starts from zero
            if (s < kElapLevels - 1) coarse_v = px_clamp(coarse, wc, hc, i3 >> 1, j3 >> 1);     // round((i - 0.5) / 2) == i / 2 for i >= 0
            if (!__builtin_isfinite(v)) v = coarse_v;               // :113-114 with the initialisation of :204-222
            const unsigned holes9 = (unsigned)__builtin_amdgcn_ballot_w64(valid && __builtin_isnan(xv));
            if (holes9 != 0u) {
                float V[9];
#pragma unroll
                for (int e = 0; e < 9; ++e) V[e] = lane_read(v, e);
#pragma unroll 1
                for (int it = 0; it < 3; ++it) {
                    float mx = 0.0f;
#pragma unroll
                    for (int e = 0; e < 9; ++e) {
                        const int ei = e % 3, ej = e / 3;
                        if ((holes9 >> e) & 1u) {                   // (uniform)
                            const float self = V[e];
                            const float right = (ei < 2 && ei < w - 1) ? V[e + 1 < 9 ? e + 1 : e] : self;
                            const float down = (ej < 2 && ej < h - 1) ? V[e + 3 < 9 ? e + 3 : e] : self;
                            const float left = ei > 0 ? V[e > 0 ? e - 1 : e] : self;
                            const float up = ej > 0 ? V[e > 2 ? e - 3 : e] : self;
                            const float lap = -4 * self + right + down + left + up;
                            const float res = self + 0.4f * lap;
                            const float update = __builtin_fabsf(self - res);
                            if (update > mx) mx = update;
                            V[e] = res;
                        }
                    }
                    if ((double)lane_read(mx, 0) < 1e-10) break;     // :66-67
                }
#pragma unroll
                for (int e = 0; e < 9; ++e) v = (lane == e) ? V[e] : v;
            }
            float *dst = sm.ver[c][0] + o;
            if (valid) dst[q] = v;
            coarse = dst;
            wc = w;
            hc = h;
            wave_sync();
            gs.mark(6);
        }
    }
#pragma unroll
    for (int s = 1; s >= 0; --s) {
        const int w = ws[s], h = hs[s], o = off[s];
        const float *x = in + o;
        if (w * h == 1) {
            // a single pixel: its four neighbours clamp to itself; the three sweeps run in registers
            if (lane == 0) {
                float v = x[0];
                if (!__builtin_isfinite(v)) {
                    v = (s < kElapLevels - 1) ? coarse[0] : 0.0f;
                    for (int it = 0; it < 3; ++it) {
                        const float lap = -4 * v + v + v + v + v;
                        const float nw = v + 0.4f * lap;
                        const float update = __builtin_fabsf(v - nw);
                        v = nw;
                        if ((double)update < 1e-10) break;
                    }
                }
                fin[o] = v;
            }
            coarse = fin + o;
            wave_sync();
            continue;
        }
        for (int q = lane; q < w * h; q += 64) {                    // :113-114 with the initialisation of :204-222
            float v = x[q];
            float coarse_v = 0.0f;                                  // coarsest level starts from zero
            if (s < kElapLevels - 1)                                // round((i - 0.5) / 2) == i / 2 for i >= 0
                coarse_v = px_clamp(coarse, ws[s + 1], hs[s + 1], (q % w) >> 1, (q / w) >> 1);
            if (!__builtin_isfinite(v)) v = coarse_v;
            sm.ver[c][0][o + q] = v; sm.ver[c][1][o + q] = v; sm.ver[c][2][o + q] = v; sm.ver[c][3][o + q] = v;
        }
        const bool lane_on = g < 3 && i < w;
        const int ic = lane_on ? i : 0;
        unsigned holes = 0;                                         // bit j: pixel (i, j) of this lane's column is a hole
        if (lane_on) {
            const int hmax = (kPatchMax + (1 << s) - 1) >> s;       // level s of an at most 11-row window (constant once s is unrolled)
#pragma unroll
            for (int j = 0; j < hmax; ++j) {                        // clamped row: the loads do not depend on h
                const float xv = x[(j < h ? j : h - 1) * w + i];
                holes |= (j < h && __builtin_isnan(xv)) ? (1u << j) : 0u;
            }
        }
        // the column to the left (lane i-1 of the same 16-lane group); row_shr:1, zero at the row start
        const unsigned holes_left = (unsigned)__builtin_amdgcn_update_dpp(0, (int)holes, 0x111, 0xf, 0xf, true);
        wave_sync();
        gs.mark(4);
        if (__builtin_amdgcn_ballot_w64(holes != 0u) == 0ull) {     // no hole at this level: the sweeps change nothing
            coarse = sm.ver[c][0] + o;
            continue;
        }
        const float *y0 = sm.ver[c][0] + o;                         // initial state: the value of every non-hole
        const float *yo = sm.ver[c][g < 3 ? g : 0] + o;             // state before this group's sweep
        float *yn = sm.ver[c][g < 3 ? g + 1 : 1] + o;               // state after it
        // Group g works on diagonal st - 3g.  What a pixel needs from the previous sweep (itself, right, lower
        // neighbour) was stored at least two steps ago and is loaded one step ahead; what it needs from its own
        // sweep was computed in the previous step: the upper neighbour by this lane (register), the left one by
        // lane i-1 (DPP).  Neighbours that are not holes keep their initial value in every version.
        int j = -3 * g - i;
        // per-lane constants of the loop: element offsets of the right / lower / left neighbour, and running state:
        // `base` = index of the row being prefetched (clamped into the level), hole bits shifted so that bit 0 is row j
        const int roff = (ic < w - 1) ? 1 : 0, loff = (ic > 0) ? 1 : 0;
        int base = ic;                                                // row clamp(j, 0, h-1) = 0 at the start (j <= 0)
        unsigned hs = lane_on ? holes : 0u, hl = holes_left;
        // Two steps per trip, written out: what a step loads for the next one (`n`) is what the step after it works on (`c`), so
        // the values change hands by their NAMES and not by register moves (seven per step of a loop the compiler would not
        // unroll: it contains the wavefront barrier).  A step past the last diagonal finds no hole bit left and changes nothing,
        // so an odd count is rounded up.
        struct SweepRow { float self, right, down, sleft; };
        SweepRow ra = { yo[base], yo[base + roff], yo[base + ((h > 1) ? w : 0)], y0[base - loff] }, rb = ra;
        float self_prev = 0.0f, nw = 0.0f, mx = 0.0f;
        bool up_hole = false;
        const int steps = w + h - 1 + 6;
        auto sweep_step = [&](const SweepRow &cr, SweepRow &nr) {
            const float left_nw = __builtin_amdgcn_update_dpp(0.0f, nw, 0x111, 0xf, 0xf, true);
            const int cur = base;                                     // == j * w + i whenever row j is inside the level
            const int jn = j + 1;
            base += (jn >= 1 && jn <= h - 1) ? w : 0;                 // next row, clamped
            const float self = cr.self, right = cr.right, down = cr.down, sleft = cr.sleft;
            nr.self = yo[base];
            nr.right = yo[base + roff];
            nr.down = yo[base + ((jn < h - 1) ? w : 0)];
            nr.sleft = y0[base - loff];
            const bool row_on = j >= 0;                               // rows >= h have no hole bit left
            const bool self_hole = row_on && (hs & 1u);
            if (self_hole) {
                // getpixel_1 clamps to the pixel itself, not yet updated in this sweep
                const float left = (i > 0) ? ((hl & 1u) ? left_nw : sleft) : self;
                const float up = (j > 0) ? (up_hole ? nw : self_prev) : self;
                const float lap = -4 * self + right + down + left + up;
                const float res = self + 0.4f * lap;
                const float update = __builtin_fabsf(self - res);
                if (update > mx) mx = update;
                yn[cur] = res;
                nw = res;
            }
            up_hole = self_hole;
            hs = row_on ? hs >> 1 : hs;
            hl = row_on ? hl >> 1 : hl;
            self_prev = self;
            j = jn;
            wave_sync();
        };
#pragma unroll 1
        for (int st = 0; st < steps; st += 2) {
            sweep_step(ra, rb);
            sweep_step(rb, ra);
        }
        // max over each 16-lane group: four row shifts accumulate it in the group's last lane (updates are >= 0, and
        // a shifted-in 0 at the row start is neutral)
        {
            float o1 = __builtin_amdgcn_update_dpp(0.0f, mx, 0x111, 0xf, 0xf, true); mx = o1 > mx ? o1 : mx;    // row_shr:1
            float o2 = __builtin_amdgcn_update_dpp(0.0f, mx, 0x112, 0xf, 0xf, true); mx = o2 > mx ? o2 : mx;    // row_shr:2
            float o4 = __builtin_amdgcn_update_dpp(0.0f, mx, 0x114, 0xf, 0xf, true); mx = o4 > mx ? o4 : mx;    // row_shr:4
            float o8 = __builtin_amdgcn_update_dpp(0.0f, mx, 0x118, 0xf, 0xf, true); mx = o8 > mx ? o8 : mx;    // row_shr:8
        }
        const float m1 = lane_read(mx, 15), m2 = lane_read(mx, 31);
        const int kfin = ((double)m1 < 1e-10) ? 1 : (((double)m2 < 1e-10) ? 2 : 3);     // :66-67
        coarse = sm.ver[c][kfin] + o;                               // the version the reference stops at
        gs.mark(s == 0 ? 5 : 6);
    }
    if (lane == 0) sm.res[c] = (int)(coarse - sm.ver[c][0]);        // where this channel's level-0 solution lies
    }                                                               // (on)
    __syncthreads();
    u1 = 0.0f;
    u2 = 0.0f;
    if (on && t < pw * ph) {
        u1 = sm.ver[0][0][sm.res[0] + t];
        u2 = sm.ver[1][0][sm.res[1] + t];
    }
    gs.mark(7);
}

// ---------------------------------------------------------------------------------------------
// of_estimation (TV-l2 coupled) on the window with the state in registers / LDS; same arithmetic
// as local_tvl1_patch_kernel (local_kernels.hpp).  Differences, all invisible in u and the energy:
// the duals never go to HBM (the reference zeroes them in the window at every call), and the
// divergence statements that land outside the window are skipped: they only ever
// write image-border entries, and a window that contains such an entry recomputes it before reading
// it.  The entries the window leaves unwritten -- first row, first / last column, unless they lie on
// the image border -- keep their HBM value, as in the reference.
// Thread 0 returns the window energy; every thread returns its new (u1, u2) in place.
// ---------------------------------------------------------------------------------------------
// weight != nullptr: the weighted functional, guided_tvl2coupled_w (src/tvl2w_model.cpp:254-455): l_t is scaled per
// pixel by weight[row] * weight[column] (:379), lambda is divided by weight[w_radio + 1]^2 with the radius in force
// during the call (src/energy_model.cpp:744-745), and the data energy is weighted (:227, a double product).
// TWIN (grow_twin_kernel.hpp): up to kTwinMax windows are minimised side by side by the 128-thread slots of one workgroup.  Every
// workgroup barrier below is then reached by all slots the same number of times: the iteration loop runs while ANY slot goes on (each
// thread reads the other slots' maxima from their LDS blocks to know), a slot that is idle (`on` = false) or has stopped only keeps the
// barriers.  tw.defer: the window's results are not stored (the caller commits them later from the registers returned in tw.d1 / d2).
// Without TWIN the function is what it was: t = threadIdx.x, one window, its own loop.
constexpr int kTwinMax = 4;     // windows a workgroup of the twin kernel computes side by side, at most
struct TwinArgs {
    int t;                      // thread index within the slot's 128 threads
    bool on, defer;             // this slot has a window in this step | its results are held back
    int self;                   // this thread's slot
    unsigned on_mask;           // bit q: slot q has a window in this step
    GrowShared *all;            // the slots' LDS blocks
    float d1, d2;               // out (defer): the thread's two divergence entries
};
template <bool TWIN = false>
__device__ __forceinline__ float window_solve(GrowShared &sm, float *base, size_t pe, int nx, int ny, int ii, int ij, int ei,
                                              int ej, int max_iter_patch, float &u1, float &u2, const float *weight, int iiw,
                                              int ijw, int wr_call, int shadow, GrowStamp &gs, TwinArgs *tw = nullptr)
{
    const int pw = ei - ii, ph = ej - ij, t = TWIN ? tw->t : (int)threadIdx.x;
    const bool on = !TWIN || tw->on;
    const bool act = on && t < pw * ph;
    const int k = ii + (act ? t % pw : 0), l = ij + (act ? t / pw : 0);
    const int p = l * nx + k;
    const float theta = 0.3f, tau = 0.125f, tol_OF = 0.01f;                       // src/energy_model.cpp:704-709
    float lambda = 40.0f, gwy = 1.0f, gwx = 1.0f;
    if (weight) {
        const float central = weight[wr_call + 1];
        lambda = lambda / (central * central);
        gwy = weight[l - ij + ijw];
        gwx = weight[k - ii + iiw];
    }
    const float l_t = weight ? lambda * theta * gwy * gwx : lambda * theta;
    const float *I0 = base + L_I0 * pe, *I1 = base + L_I1 * pe, *I1x = base + L_I1X * pe, *I1y = base + L_I1Y * pe;
    float *D1 = base + L_DIV1 * pe, *D2 = base + L_DIV2 * pe;

    // which divergence statement of src/utils.cpp:76-108 writes this thread's own entry
    const bool k_in = k > ii && k < ei - 1, l_in = l > ij && l < ej - 1;
    const bool first_row = l == ij, last_row = l == ej - 1, first_col = k == ii, last_col = k == ei - 1;
    const bool corner = (first_row || last_row) && (first_col || last_col);
    const bool fresh = act && ((k_in && l_in) || (k_in && last_row) || corner || (k_in && first_row && ij == 0) ||
                               (l_in && first_col && ii == 0) || (l_in && last_col && ei == nx));

    // Operand selectors of the divergence (loop invariant).  Every statement of divergence_patch has the form
    // (P + Q) [+ R] with P in {x1, -left}, Q in {-left, x2, -up}, R in {x2, -up, x2 - up}; a - b is a + (-b) and
    // -a + b is b + (-a) bit for bit, so selecting the operands replaces ten divergent branches.
    const bool interior = k_in && l_in;
    const bool selP_left = last_col && (corner || l_in);                               // -v1[p-1] + ...
    const bool selQ_left = k_in;                                                       // x1 - left
    const bool selQ_up = corner && last_row;                                           // ... - v2[p-nx] as 2nd operand
    const bool selR_none = corner;
    const bool selR_x2 = k_in && first_row;
    const int lidx = first_col ? t : t - 1, tidx = first_row ? t : t - pw;             // own slot where the operand is unused
    const int ridx = last_col ? t : t + 1, didx = last_row ? t : t + pw;

    float x11 = 0, x12 = 0, x21 = 0, x22 = 0, rc = 0, wx = 0, wy = 0, gr = 0, v1 = 0, v2 = 0, ub1 = 0, ub2 = 0, d1 = 0, d2 = 0,
          i0 = 0;
    double rgrad = 0.0;
    if (act) {
        d1 = D1[p];
        d2 = D2[p];
        i0 = I0[p];
        const float uu = (float)k + u1, vv = (float)l + u2;                       // src/bicubic_interpolation.c:288-289
        const BicubicTaps tp = bicubic_taps(uu, vv, nx, ny);                      // border_out = false
        const bool anyc = wave_any(tp.clamped);
        const float w0 = bicubic_eval_any(I1, tp, nx, anyc);
        wx = bicubic_eval_any(I1x, tp, nx, anyc);
        wy = bicubic_eval_any(I1y, tp, nx, anyc);
        gr = grad_sq(wx, wy);
        rc = rho_const(w0, wx, wy, u1, u2, i0);
        rgrad = recip_f64(gr);
        ub1 = u1;
        ub2 = u2;
    }
    const double rtheta = 1.0 / (double)theta;
    sm.a[t] = ub1;
    sm.b[t] = ub2;
    __syncthreads();
    gs.mark(8);
    int it = 0, it_o[kTwinMax] = { 0, 0, 0, 0 };
    float err_D = __builtin_inff();
    // The per-pixel steps are the ones of the global solver (faldoi_device.hpp): same expressions as
    // src/tvl2_model.cpp:364-427, with the exact-quotient sequences instead of the IEEE divide.
    bool go = on && max_iter_patch > 0;                                           // src/tvl2_model.cpp:359 (err_D starts at infinity)
    unsigned go_o = (TWIN && max_iter_patch > 0) ? (tw->on_mask & ~(1u << tw->self)) : 0u;       // the other slots that go on
    while (go || go_o) {
        it += go ? 1 : 0;
        if (TWIN) {
#pragma unroll
            for (int q = 0; q < kTwinMax; ++q) it_o[q] += (int)((go_o >> q) & 1u);
        }
        if (act && go) {
            threshold_step(rc, wx, wy, gr, rgrad, u1, u2, l_t, v1, v2);           // :364-391
            // forward_gradient_patch (src/utils.cpp:175-220) + tvl2coupled_getD (:82-115)
            const float ar = sm.a[ridx], ad = sm.a[didx], br = sm.b[ridx], bd = sm.b[didx];
            const float u1x = !last_col ? ar - ub1 : 0.0f, u1y = !last_row ? ad - ub1 : 0.0f;
            const float u2x = !last_col ? br - ub2 : 0.0f, u2y = !last_row ? bd - ub2 : 0.0f;
            dual_step(x11, x12, x21, x22, u1x, u1y, u2x, u2y, tau);
        }
        sm.x[0][t] = x11; sm.x[1][t] = x12; sm.x[2][t] = x21; sm.x[3][t] = x22;
        __syncthreads();
        float uNr = 0.0f;
        if (act && go) {
            if (fresh) {                                                          // divergence_patch, src/utils.cpp:63-113
                const float nl11 = -sm.x[0][lidx], nl21 = -sm.x[2][lidx], nt12 = -sm.x[1][tidx], nt22 = -sm.x[3][tidx];
                const float s1 = (selP_left ? nl11 : x11) + (selQ_left ? nl11 : (selQ_up ? nt12 : x12));
                const float s2 = (selP_left ? nl21 : x21) + (selQ_left ? nl21 : (selQ_up ? nt22 : x22));
                const float r1 = interior ? x12 + nt12 : (selR_x2 ? x12 : nt12);
                const float r2 = interior ? x22 + nt22 : (selR_x2 ? x22 : nt22);
                d1 = selR_none ? s1 : s1 + r1;
                d2 = selR_none ? s2 : s2 + r2;
            }
            const float uN = primal_step(u1, u2, ub1, ub2, v1, v2, d1, d2, theta, rtheta, tau);   // :122-166, :420-427
            uNr = uN > 0.0f ? uN : 0.0f;                                          // NaN never wins, as in the reference
        }
        // max over the window: inside each wavefront by six DPP steps (a `__shfl_xor` butterfly is six dependent trips through the
        // LDS crossbar per iteration), then one LDS word per wavefront
        const float wmax = wave_max_nonneg(uNr);
        if ((t & 63) == 0 && go) sm.err[2 * (it & 1) + (t >> 6)] = __float_as_int(wmax);
        sm.a[t] = ub1;
        sm.b[t] = ub2;
        __syncthreads();
        if (go) {
            const float e0 = __int_as_float(sm.err[2 * (it & 1)]), e1 = __int_as_float(sm.err[2 * (it & 1) + 1]);
            err_D = e1 > e0 ? e1 : e0;
            go = err_D > tol_OF * tol_OF && it < max_iter_patch;
        }
        if (TWIN && go_o) {                                                       // the other slots' decisions, from their own maxima
#pragma unroll
            for (int q = 0; q < kTwinMax; ++q)
                if ((go_o >> q) & 1u) {
                    const int *e = tw->all[q].err;
                    const float o0 = __int_as_float(e[2 * (it_o[q] & 1)]), o1 = __int_as_float(e[2 * (it_o[q] & 1) + 1]);
                    const float err_o = o1 > o0 ? o1 : o0;
                    if (!(err_o > tol_OF * tol_OF && it_o[q] < max_iter_patch)) go_o &= ~(1u << q);
                }
        }
    }
    __syncthreads();                                     // every read of ubar is done before u overwrites it
    gs.mark(9);

    // ---- eval_tvl2coupled (src/tvl2_model.cpp:174-243) ----
    sm.a[t] = u1;
    sm.b[t] = u2;
    __syncthreads();
    float term = 0.0f;
    if (TWIN) { tw->d1 = d1; tw->d2 = d2; }
    if (act) {
        if (!(TWIN && tw->defer)) {
            base[L_U1 * pe + p] = u1;
            base[L_U2 * pe + p] = u2;
            D1[p] = d1;
            D2[p] = d2;
        }
        if (shadow) {
            // Partitioned runs change the domain between growings, so what is a border entry now may be an interior
            // (stale-read) entry later: the duals are kept in HBM and the statements of divergence_patch that land on
            // the DOMAIN's row 0 / column 0 / column nx-1 outside the window (src/utils.cpp:90,98,101) are carried out.
            // They read duals this call does not touch, so once, after the last iteration, is the same as every time.
            float *X11 = base + L_X11 * pe, *X12 = base + L_X12 * pe, *X21 = base + L_X21 * pe, *X22 = base + L_X22 * pe;
            X11[p] = x11; X12[p] = x12; X21[p] = x21; X22[p] = x22;
            if (k_in && first_row && ij > 0) {                                    // :90  div[k] = v1[k] - v1[k-1] + v2[k]
                D1[k] = X11[k] - X11[k - 1] + X12[k];
                D2[k] = X21[k] - X21[k - 1] + X22[k];
            }
            if (l_in && first_col && ii > 0) {                                    // :98  p1 = l*nx
                const int p1 = l * nx;
                D1[p1] = X11[p1] + X12[p1] - X12[p1 - nx];
                D2[p1] = X21[p1] + X22[p1] - X22[p1 - nx];
            }
            if (l_in && last_col && ei < nx) {                                    // :101 p2 = (l+1)*nx - 1
                const int p2 = (l + 1) * nx - 1;
                D1[p2] = -X11[p2 - 1] + X12[p2] - X12[p2 - nx];
                D2[p2] = -X21[p2 - 1] + X22[p2] - X22[p2 - nx];
            }
        }
        const float ar = sm.a[ridx], ad = sm.a[didx], br = sm.b[ridx], bd = sm.b[didx];
        const float u1x = !last_col ? ar - u1 : 0.0f, u1y = !last_row ? ad - u1 : 0.0f;
        const float u2x = !last_col ? br - u2 : 0.0f, u2y = !last_row ? bd - u2 : 0.0f;
        const BicubicTaps tp = bicubic_taps((float)k + u1, (float)l + u2, nx, ny);
        const float iw = bicubic_eval_any(I1, tp, nx, wave_any(tp.clamped));
        const float dt = weight ? (float)((double)lambda * __builtin_fabs((double)(iw - i0)) * (double)gwy * (double)gwx)
                                : lambda * __builtin_fabsf(iw - i0);
        const float dc = (1 / (2 * theta)) * ((u1 - v1) * (u1 - v1) + (u2 - v2) * (u2 - v2));
        const float g = __builtin_sqrtf(u1x * u1x + u1y * u1y + u2x * u2x + u2y * u2y);
        term = dc + dt + g;
    }
    sm.t[t] = term;
    __syncthreads();
    float ener = 0.0f;
    if (t == 0 && on) {                                  // the reference accumulates in row-major order, in fp32
        const int m = pw * ph;
        float4 v[(kPatchMax * kPatchMax + 3) / 4];       // all LDS reads in flight first, then the ordered sum
        const float4 *t4 = reinterpret_cast<const float4 *>(sm.t);
#pragma unroll
        for (int q = 0; q < (kPatchMax * kPatchMax + 3) / 4; ++q) v[q] = t4[q];
        if (m == kPatchMax * kPatchMax) {                // the interior case: 121 plain additions
#pragma unroll
            for (int q = 0; q < kPatchMax * kPatchMax / 4; ++q) { ener += v[q].x; ener += v[q].y; ener += v[q].z; ener += v[q].w; }
            ener += v[kPatchMax * kPatchMax / 4].x;
        } else {
#pragma unroll
            for (int q = 0; q < (kPatchMax * kPatchMax + 3) / 4; ++q) {
                if (4 * q + 0 < m) ener += v[q].x;
                if (4 * q + 1 < m) ener += v[q].y;
                if (4 * q + 2 < m) ener += v[q].z;
                if (4 * q + 3 < m) ener += v[q].w;
            }
        }
        ener = (float)((double)ener / (m * 1.0));
    }
    gs.mark(10);
    return ener;
}

// ---------------------------------------------------------------------------------------------
// The candidate queue on the device: the element moves of grow_heap.hpp (the definition, shared with
// the host), carried out by the 64 lanes of the FIRST wavefront together.  A heap walk by one lane is a
// chain of dependent round trips and, above all, ~1000 instructions of a wavefront that issues one
// instruction per ~5 cycles; with the lanes side by side:
//   pop : the 62 descendants of the hole down to depth 5 are loaded by lanes 0..61 in ONE instruction
//         (lane L holds the node at depth d = floor(log2(L+2)), offset L+2-2^d); the walk compares two
//         `readlane`d energies per level, marks the lane of the winning child, and the marked lanes store
//         their element into their parent's slot -- five levels per round trip;
//   push: lane q loads the q-th ancestor of the new slot; a ballot of `ancestor.sim > value.sim` gives the
//         number m of leading ancestors that move down (count of trailing ones); lanes 0..m-1 store one
//         slot lower, lane 0 drops the value into ancestor m-1's slot.
// Every function below is called by all lanes of wavefront 0 with identical arguments.
// Elements travel as native 4-vectors {sim, u, v, bits(ij)} (selecting between structs, or HIP's
// union-based float4, goes through memory).
// ---------------------------------------------------------------------------------------------
typedef float cand4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) cand4 *heap_ptr;      // the queue lives in global memory: plain global loads
typedef __attribute__((address_space(3))) cand4 *heap_lds_ptr;

// The queue as a growing kernel sees it: while the kernel runs, the first `ntop` elements (the top levels of the binary
// heap: 2^L - 1 of them) live in LDS -- every pop starts at the root and every sift-up ends near it, so with the levels that
// fit (L = 13 when the CU holds one growing, 10 with four) a pop makes one or two trips to global memory instead of four to
// six.  The kernel copies them in when it starts and back when it leaves; the element moves are those of grow_heap.hpp
// whichever memory an element happens to be in.
struct HeapRef {
    heap_ptr g;
    heap_lds_ptr top;
    int ntop;                         // 2^L - 1 (whole levels) or 0
    __device__ __forceinline__ cand4 ld(int i) const
    {
        cand4 v;
        if (i < ntop) v = top[i];
        else v = g[i];
        return v;
    }
    __device__ __forceinline__ void st(int i, const cand4 v) const
    {
        if (i < ntop) top[i] = v;
        else g[i] = v;
    }
};
__device__ __forceinline__ HeapRef heap_global_only(Candidate *heap) { return HeapRef{ (heap_ptr)heap, (heap_lds_ptr) nullptr, 0 }; }
// All threads of the workgroup: the top of a queue of `size` elements into LDS / back (barriers by the caller)
__device__ __forceinline__ void heap_top_in(const HeapRef &h, int size)
{
    const int n = size < h.ntop ? size : h.ntop;
    for (int k = threadIdx.x; k < n; k += blockDim.x) h.top[k] = h.g[k];
}
__device__ __forceinline__ void heap_top_out(const HeapRef &h, int size)
{
    const int n = size < h.ntop ? size : h.ntop;
    for (int k = threadIdx.x; k < n; k += blockDim.x) h.g[k] = h.top[k];
}

__device__ __forceinline__ void heap_sift_up_wave(const HeapRef &h, int hole, const cand4 value)
{
    const int lane = threadIdx.x & 63;
    while (hole > 0) {
        const bool have = lane < 24 && ((hole + 1) >> (lane + 1)) > 0;        // lane q <-> q-th ancestor
        const int aidx = have ? ((hole + 1) >> (lane + 1)) - 1 : 0;
        // (both memories asked at once, into registers of their own: HeapRef::ld's two-sided form waits for the global answer
        // before it asks LDS)
        // -- every lane asks both, the one that does not hold its element for a slot that exists all the same: the queue's array
        // is allocated from slot 0 on, its top merely stale while the kernel runs)
        const bool in_top = aidx < h.ntop;
        const cand4 below = h.g[aidx];
        cand4 mine = below;
        if (h.ntop > 0) {
            const cand4 above = h.top[in_top ? aidx : 0];
            mine.x = in_top ? above.x : below.x;
            mine.y = in_top ? above.y : below.y;
            mine.z = in_top ? above.z : below.z;
            mine.w = in_top ? above.w : below.w;
        }
        const unsigned long long down = __builtin_amdgcn_ballot_w64(have && mine.x > value.x);
        const int m = __builtin_ctzll(~down);                                 // leading ancestors that move down
        if (lane < m) h.st(((hole + 1) >> lane) - 1, mine);                   // into the slot of ancestor lane-1 (the hole for lane 0)
        if (m == 0) break;
        hole = ((hole + 1) >> m) - 1;
        if (m < 24) break;
    }
    if (lane == 0) h.st(hole, value);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // later loads by OTHER lanes must see these stores
}

__device__ __forceinline__ void heap_push_wave(const HeapRef &h, int &size, int px, int py, float u, float v, float sim)
{
    heap_sift_up_wave(h, size, cand4{ sim, u, v, __uint_as_float((unsigned)px | ((unsigned)py << 16)) });
    ++size;
}

// One round of a pop's descent (std::__adjust_heap), at most six levels.  A pop is bounded by the instructions ONE wavefront can
// issue (tools/heap_cycles.py: the same ~5.5 k cycles whether the queue lies in LDS or in global memory), so the round is built
// to need few of them and no traffic between the register files inside the walk:
//   * lane L holds node L of the hole's subtree in heap numbering (0: the hole itself, children of L: 2L+1, 2L+2; depths 0..5 in
//     lanes 0..62), a second register holds depth 6 (lanes 0..63).  Besides its element, every lane requests the ENERGIES OF ITS
//     NODE'S TWO CHILDREN (two more 4-byte loads, in flight with the others: one round trip per round);
//   * ONE vector comparison then decides, for all 63 nodes at once, which child std::__adjust_heap would take there
//     (`comp(right, left)`: the float comparison as written), and lands in a scalar register pair as a bit mask; a second mask
//     says which nodes have two children at all;
//   * the walk is scalar bit arithmetic on those masks -- rel = 2 rel + 1 + bit(rel) -- and marks the path's lanes;
//   * the marked lanes store their elements into their parents' slots.
// (Built and measured before this form: a `readlane` pair and a compare per level, ~20 instructions a level with the moves between
// the register files -- 5.5 k cycles a pop; the same with eight levels per trip, two and four more registers for depths 7 and 8 --
// slower still: the selects and the 510 elements per pop cost more than the trip they save.)
//
// The round that reaches a leaf also does std::__push_heap's part without another trip: the parent chain of the final hole IS the
// path just walked, its elements are in registers.  The lowest path element the value does not pass (`!(path.sim > value.sim)`)
// and everything above it move up, everything below it stays, and the value drops into the slot that element leaves.  Only when
// the value passes every path element of this round (or the path ends at the one node with a single child) does the caller fall
// back to the general sift-up.  Returns true when the value has been placed.  All arguments wave-uniform.
constexpr int kHeapRoundLevels = 6;
#ifdef FALDOI_HEAP_STAMPS                                            // what-if build only (tools/heap_cycles.py): where a pop's cycles go
__device__ unsigned long long g_heap_stamp[16];
__device__ __forceinline__ unsigned long long *heap_stamp_acc() { __shared__ unsigned long long acc[16]; return acc; }
#define HEAP_STAMP(k) do { const long long now_ = clock64(); if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(heap_stamp_acc() + (k), (unsigned long long)(now_ - hs_last), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); hs_last = clock64(); } while (0)
#define HEAP_STAMP_BEGIN long long hs_last = clock64()
#else
#define HEAP_STAMP(k) do { } while (0)
#define HEAP_STAMP_BEGIN do { } while (0)
#endif
typedef __attribute__((address_space(1))) const float *heap_key_ptr;
typedef __attribute__((address_space(3))) const float *heap_lds_key_ptr;

// What a lane needs to know about its node of a round's subtree (it depends on the lane alone)
struct HeapLane {
    int d;                            // depth of node `lane` under the hole, 0..5 (lane 63: unused)
    int parent;                       // its parent's node (lane 0: 0)
    bool is_left;                     // an odd node is its parent's left child
    unsigned long long line;          // the nodes from this one up to depth 1 (lanes 0 and 63: all bits -- never matched)
};
__device__ __forceinline__ HeapLane heap_lane()
{
    const int lane = threadIdx.x & 63;
    HeapLane hl;
    hl.d = 31 - __builtin_clz(lane + 1);
    hl.parent = lane ? (lane - 1) >> 1 : 0;
    hl.is_left = (lane & 1) != 0;
    unsigned long long line = 0ull;
    int x = lane;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (x >= 1) line |= 1ull << x;
        x = x >= 1 ? (x - 1) >> 1 : 0;
    }
    hl.line = (lane >= 1 && lane <= 62) ? line : ~0ull;
    return hl;
}

template <bool GLOBAL>
__device__ __forceinline__ bool heap_descend_round(const HeapRef &h, int &hole, const int half, const int last, const int nl, cand4 &value,
                                                   bool &have_value, const int single, const HeapLane &hl)
{
    HEAP_STAMP_BEGIN;
    const int lane = threadIdx.x & 63;
    const int d = hl.d;
    const int idx0 = (hole << d) + lane, idx6 = (hole << 6) + 63 + lane;
    const bool want0 = lane >= 1 && lane <= 62 && d <= nl && idx0 <= last;
    const bool six = nl >= 6 && hole + 1 <= ((last + 1) >> 6);      // uniform: the subtree reaches depth 6
    const bool dec = lane <= 62 && d < nl && idx0 < half;           // a node with two children below which the round may step
    const int kid = 2 * idx0 + 1;
    // (plain loads from the one memory the round's levels lie in -- HeapRef::ld's two-sided form makes every load wait for the one
    // before it, both sides writing the same registers; this is why ntop must be 2^L - 1: whole levels)
    const cand4 none = { 0.0f, 0.0f, 0.0f, 0.0f };
    cand4 r0, r6 = none;
    float kl, kr;
    if (GLOBAL) {
        const heap_key_ptr keys = (heap_key_ptr)h.g;
        r0 = h.g[want0 ? idx0 : last];
        kl = keys[4 * (size_t)(dec ? kid : last)];
        kr = keys[4 * (size_t)(dec ? kid + 1 : last)];
        if (six) r6 = h.g[idx6 <= last ? idx6 : last];
        if (!have_value) {                                          // the queue's last element, behind the round's own requests
            value = h.g[last + 1];
            have_value = true;
        }
    } else {
        const heap_lds_key_ptr keys = (heap_lds_key_ptr)h.top;
        r0 = h.top[want0 ? idx0 : 0];
        kl = keys[4 * (dec ? kid : 0)];
        kr = keys[4 * (dec ? kid + 1 : 0)];
        if (six) r6 = h.top[idx6 <= last ? idx6 : 0];
    }
    HEAP_STAMP(GLOBAL ? 4 : 0);                                     // addresses, loads issued
    const unsigned long long two = __builtin_amdgcn_ballot_w64(dec);
    const unsigned long long goes_right = __builtin_amdgcn_ballot_w64(!(kr > kl));      // comp(right, left) picks the left child
    HEAP_STAMP(GLOBAL ? 5 : 1);                                     // loads arrived, decisions
    // Seen from the children: node L is TAKEN if its parent has two children and goes L's way (parents are nodes 0..30: the masks'
    // low words); it is ON THE PATH if it and every node above it, up to depth 1, are taken.
    const unsigned takes_left = (unsigned)(two & ~goes_right), takes_right = (unsigned)(two & goes_right);
    const bool taken = (((hl.is_left ? takes_left : takes_right) >> hl.parent) & 1u) != 0u;
    const unsigned long long taken_mask = __builtin_amdgcn_ballot_w64(taken);
    const bool onpath = (taken_mask & hl.line) == hl.line;
    const unsigned long long pm = __builtin_amdgcn_ballot_w64(onpath);                  // (never empty: the hole has two children)
    const int deep = 63 - __builtin_clzll(pm);                      // the path's lowest node among depths 1..5
    int r = __builtin_popcountll(pm);                               // levels walked
    int c6 = -1;                                                    // the lane of r6 on the path
    if (r == 5 && ((two >> deep) & 1ull) != 0ull) {
        c6 = 2 * deep + 1 + (int)((goes_right >> deep) & 1ull) - 63;
        r = 6;
    }
    const int cur = c6 >= 0 ? (hole << 6) + 63 + c6 : (hole << r) + deep;
    HEAP_STAMP(GLOBAL ? 6 : 2);                                     // walk
    // the elements on the path move into their parents' slots -- all of them, unless the value can be placed from here
    bool placed = false, move6 = c6 >= 0, stop6 = false;
    int stop = 64;                                                  // the lane whose slot takes the value; path lanes above it stay
    if (cur >= half && cur != single && have_value) {               // a leaf: __push_heap from here, on the path in registers
        const float vkey = lane_read(value.x, 0);
        if (c6 >= 0 && !(lane_read(r6.x, c6) > vkey)) {             // the value does not pass the leaf's parent-to-be
            placed = true;
            stop6 = true;
        } else {
            const unsigned long long stay = pm & ~__builtin_amdgcn_ballot_w64(r0.x > vkey);     // path elements the value does not pass
            if (stay != 0ull) {                                     // the lowest of them (the path's lanes are in depth order)
                placed = true;
                stop = 63 - __builtin_clzll(stay);
                move6 = false;
            }
        }
    }
    if (onpath && lane <= stop) {
        if (GLOBAL) {
            h.st((idx0 - 1) >> 1, r0);                              // (the hole itself may still lie in LDS)
            if (lane == stop) h.g[idx0] = value;
        } else {
            h.top[(idx0 - 1) >> 1] = r0;
            if (lane == stop) h.top[idx0] = value;
        }
    }
    if (move6 && lane == c6) {
        if (GLOBAL) {
            h.g[(idx6 - 1) >> 1] = r6;
            if (stop6) h.g[idx6] = value;
        } else {
            h.top[(idx6 - 1) >> 1] = r6;
            if (stop6) h.top[idx6] = value;
        }
    }
    hole = cur;
    HEAP_STAMP(GLOBAL ? 7 : 3);                                     // placement, stores issued
    return placed;
}

// Removes the top (the caller has read element 0); precondition: size > 0.
__device__ __forceinline__ void heap_remove_top_wave(const HeapRef &h, int &size)
{
    const int len = __builtin_amdgcn_readfirstlane(size) - 1;      // (the lanes hold the same counters: scalar control flow from here)
    size = len;
    if (len == 0) return;
    HEAP_STAMP_BEGIN;
    // the last element: from LDS if it lies there; otherwise requested by the first round below the LDS levels, behind its own loads
    // (a request made here would be waited for before that round could issue anything)
    cand4 value = { 0.0f, 0.0f, 0.0f, 0.0f };
    bool have_value = len < h.ntop;
    if (have_value) value = h.top[len];
    const int half = (len - 1) / 2;                                 // nodes below `half` have two children
    const int last = len - 1;
    const int single = (len & 1) == 0 ? (len - 2) / 2 : -1;        // the one node with a single (left) child, if there is one
    const int top_levels = 31 - __builtin_clz(h.ntop + 1);         // levels of the queue that live in LDS (ntop = 2^L - 1, or 0)
    int hole = 0;
    bool placed = false;
    const HeapLane hl = heap_lane();
    HEAP_STAMP(8);                                                  // set-up
    while (hole < half && !placed) {
        const int in_lds = top_levels - (32 - __builtin_clz(hole + 1));        // levels below the hole that lie in LDS
        if (in_lds >= 1) placed = heap_descend_round<false>(h, hole, half, last, in_lds < kHeapRoundLevels ? in_lds : kHeapRoundLevels, value, have_value, single, hl);
        else placed = heap_descend_round<true>(h, hole, half, last, kHeapRoundLevels, value, have_value, single, hl);
    }
#ifdef FALDOI_HEAP_STAMPS
    hs_last = clock64();
#endif
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    HEAP_STAMP(9);                                                  // fence
    if (placed) return;
    HEAP_STAMP(10);
    if (!have_value) value = h.g[len];                              // (a queue that ends on the last LDS level: no round below it was made)
    if (hole == single) {
        const int child = 2 * (hole + 1);
        const cand4 c = h.ld(child - 1);
        if ((threadIdx.x & 63) == 0) h.st(hole, c);
        hole = child - 1;
    }
    heap_sift_up_wave(h, hole, value);
}

// add_neighbors (src/local_faldoi.cpp:679-727) for the pixel (i, j), all 128 threads.
// FN: which window solver the kernel is built around -- 0: TV-l2 coupled (window_solve); 1: non-local + L1 (-m 2, 3);
// 2: CSAD alone (-m 4, 5); 3: non-local + CSAD (-m 6); 4: non-local + CSAD over sqrt(|grad|^2) (-m 7).
template <int FN>
__device__ __forceinline__ void add_neighbors_dev(GrowShared &sm, NlShared *ns, const GrowDir &D, size_t pe, int nx, int ny, int i, int j,
                                                  int iteration, int wr, int max_iter_patch, const HeapRef &heap, int &heap_size,
                                                  long long &fills, long long *cyc, GrowStamp &gs)
{
    const int t = threadIdx.x;
    const int ii = (i - wr < 0) ? 0 : i - wr, ij = (j - wr < 0) ? 0 : j - wr;             // get_index_patch
    const int ei = (i + 1 + wr > nx) ? nx : i + 1 + wr, ej = (j + 1 + wr > ny) ? ny : j + 1 + wr;
    const int pw = ei - ii, ph = ej - ij;
    const bool act = t < pw * ph;
    const int p = (ij + (act ? t / pw : 0)) * nx + ii + (act ? t % pw : 0);
    bool fill = iteration == 0;
    if (!fill) {                                                                  // check_trustable_patch :639-656: each wavefront votes on
        // its part of the window, one barrier (a __syncthreads_or is two and an LDS atomic)
        const unsigned long long vote = __builtin_amdgcn_ballot_w64(act && D.trust[p] == 0);
        if ((t & 63) == 0) sm.vote[t >> 6] = vote != 0ull ? 1 : 0;
        __syncthreads();
        fill = (sm.vote[0] | sm.vote[1]) != 0;
    }
    // first wavefront (uniformly): state of the four neighbours and of the centre, read now so that the latency
    // hides behind the window computation (only lane 0 ever writes these arrays during a growing)
    // (lane q < 4 holds neighbour q's state, lane 4 the centre's: one set of loads for all five)
    const int nbx[4] = { 0, 0, 1, -1 }, nby[4] = { 1, -1, 0, 0 };                 // insert_candidates :501-505
    int nfix = 1;
    float nene = 0.0f, nsal = 1.0f;
    if (t < 64) {
        const int px = i + (t == 2 ? 1 : t == 3 ? -1 : 0), py = j + (t == 0 ? 1 : t == 1 ? -1 : 0);
        const bool inside = px >= 0 && px < nx && py >= 0 && py < ny;
        const int g = inside ? py * nx + px : j * nx + i;
        const int f = D.fixed[g];
        nfix = inside ? f : 1;
        nene = D.ene[g];
        if (D.sal) nsal = D.sal[g];
    }
    float u1 = 0.0f, u2 = 0.0f;
    gs.mark(1);
    const long long c0 = clock64();
    if (fill) {
        harmonic_fill_window(sm, D.out, pe, D.fixed, nx, ii, ij, pw, ph, u1, u2, gs, t);
        if (t == 0) ++fills;
    } else if (act) {
        u1 = D.planes[L_U1 * pe + p];
        u2 = D.planes[L_U2 * pe + p];
    }
    const long long c1 = clock64();
    const int iiw = (i - wr < 0) ? -(i - wr) : 0, ijw = (j - wr < 0) ? -(j - wr) : 0;    // get_relative_index_weight :551-557
    float ener;
    if constexpr (FN == 0)
        ener = window_solve(sm, D.planes, pe, nx, ny, ii, ij, ei, ej, max_iter_patch, u1, u2, D.weight, iiw, ijw, wr, D.shadow, gs);
    else if constexpr (FN == 1)
        ener = window_solve_nl<true, kDataL1>(sm, *ns, D.planes, D.duals, D.wp, D.planes, pe, nx, ny, ii, ij, ei, ej, max_iter_patch,
                                              u1, u2, D.weight, iiw, ijw, wr, D.method);
    else if constexpr (FN == 2)
        ener = window_solve_nl<false, kDataCsad>(sm, *ns, D.planes, D.duals, D.wp, D.planes, pe, nx, ny, ii, ij, ei, ej,
                                                 max_iter_patch, u1, u2, D.weight, iiw, ijw, wr, D.method);
    else if constexpr (FN == 3)
        ener = window_solve_nl<true, kDataCsad>(sm, *ns, D.planes, D.duals, D.wp, D.planes, pe, nx, ny, ii, ij, ei, ej,
                                                max_iter_patch, u1, u2, D.weight, iiw, ijw, wr, D.method);
    else
        ener = window_solve_nl<true, kDataCsadSqrt>(sm, *ns, D.planes, D.duals, D.wp, D.planes, pe, nx, ny, ii, ij, ei, ej,
                                                    max_iter_patch, u1, u2, D.weight, iiw, ijw, wr, D.method);
    const long long c2 = clock64();
    if (t < 64) {                                                                 // insert_candidates :497-537
        const float ener0 = lane_read(ener, 0);                                   // lane 0 summed the energy
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int px = i + nbx[q], py = j + nby[q];
            const float new_ener = ener0 * lane_read(nsal, q);
            if (!__builtin_amdgcn_readlane(nfix, q) && new_ener < lane_read(nene, q)) {       // outside the image: nfix = 1 (wave-uniform test)
                if (t == 0) D.ene[py * nx + px] = ener0;
                const int wq = (py - ij) * pw + (px - ii);                        // the neighbour lies in the window
                if (heap_size < D.heap_cap) heap_push_wave(heap, heap_size, px, py, sm.a[wq], sm.b[wq], new_ener);
                else if (t == 0) D.ctl->overflow = 1;
            }
        }
        const int g = j * nx + i, wq = (j - ij) * pw + (i - ii);                  // :718-726
        if (t == 0 && lane_read(nene, 4) > ener0) {
            D.out[g] = sm.a[wq];
            D.out[pe + g] = sm.b[wq];
            D.ene[g] = ener0;
        }
    }
    gs.mark(11);
    if (t == 0) {
        cyc[0] += clock64() - c2;
        cyc[1] += c1 - c0;
        cyc[2] += c2 - c1;
    }
    __syncthreads();
}

// mode 0: the seed pass of insert_initial_seeds (:764-783); mode 1: local_growing (:899-937).
template <int FN>
__device__ __forceinline__ void grow_body(GrowShared &sm, NlShared *ns, const GrowDir *__restrict__ dirs, int mode, int iteration,
                                          int wr, int max_iter_patch, int budget, int ntop)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char grow_heap_top[];     // ntop queue elements (dynamic LDS, host-sized)
    const GrowDir D = dirs[blockIdx.x];
    const HeapRef heap{ (heap_ptr)D.heap, (heap_lds_ptr)(reinterpret_cast<cand4 *>(grow_heap_top)), ntop };
    heap_top_in(heap, D.ctl->heap_size);
    __syncthreads();
    const int t = threadIdx.x;
    const int nx = D.nx, ny = D.ny;
    const size_t pe = D.pe;
    int heap_size = 0, seed_pos = 0;
    long long pops = 0, solves = 0, fills = 0;
    long long cyc[3] = { 0, 0, 0 };
    if (t < 64) {                                   // the first wavefront runs the queue; its lanes hold the same counters
        heap_size = D.ctl->heap_size;
        seed_pos = D.ctl->seed_pos;
    }
    int done = 0;
    GrowStamp gs{ D.detail != 0 && blockIdx.x == 0 && t == 0 && mode == 1, 0 };
    gs.start();
    for (int step = 0; step < budget; ++step) {
        if (t < 64) {
            int have = 0, g = 0;
            const long long c0 = clock64();
            if (mode == 0) {
                if (seed_pos < D.nseeds) {
                    g = D.seeds[seed_pos++];
                    if (t == 0) {
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
                    if (t == 0) {
                        D.fixed[g] = 1;                                           // :926-932
                        D.out[g] = c.y;
                        D.out[pe + g] = c.z;
                        D.ene[g] = c.x;
                    }
                    have = 1;
                    break;
                }
            }
            if (t == 0) {
                cyc[0] += clock64() - c0;
                sm.cur[0] = have;
                sm.cur[1] = g % nx;
                sm.cur[2] = g / nx;
            }
        }
        __syncthreads();
        if (!sm.cur[0]) { done = 1; break; }
        gs.mark(0);
        const int i = sm.cur[1], j = sm.cur[2];
        add_neighbors_dev<FN>(sm, ns, D, pe, nx, ny, i, j, iteration, mode == 0 ? 1 : wr, max_iter_patch, heap, heap_size, fills, cyc, gs);
        if (t == 0) {
            ++solves;
            if (mode == 0) {                                                      // :778-780
                const int g = j * nx + i;
                D.out[g] = __builtin_nanf("");
                D.out[pe + g] = __builtin_nanf("");
                D.fixed[g] = 0;
            }
        }
    }
    if (t == 0) sm.cur[3] = heap_size;
    __syncthreads();                                // (every queue operation of the first wavefront is behind its fence)
    heap_top_out(heap, sm.cur[3]);
    if (t == 0) {
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

// Two builds of the same body.  A growing is latency-bound (one instruction per ~5 cycles per wavefront), so a CU
// gains from more resident growings as long as they fit: the default build uses ~190 VGPRs (2 wavefronts per SIMD,
// 4 growings per CU); the dense build is held to 128 VGPRs (a few spills, ~8 % slower per growing) and fits 8 per
// CU.  Measured on 256x192 pairs: 512 pairs 26.7 M windows/s with the default build, 1024 pairs 28.6 M (default)
// vs 35.7 M (dense).  The host picks by the number of growings.
__global__ __launch_bounds__(kPatchThreads) void grow_kernel(const GrowDir *__restrict__ dirs, int mode, int iteration, int wr,
                                                           int max_iter_patch, int budget, int ntop)
{
    __shared__ GrowShared sm;
    grow_body<0>(sm, nullptr, dirs, mode, iteration, wr, max_iter_patch, budget, ntop);
}

__global__ __launch_bounds__(kPatchThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void grow_kernel_dense(const GrowDir *__restrict__ dirs, int mode, int iteration, int wr, int max_iter_patch, int budget, int ntop)
{
    __shared__ GrowShared sm;
    grow_body<0>(sm, nullptr, dirs, mode, iteration, wr, max_iter_patch, budget, ntop);
}

// The same growing around the window solvers of the other functionals (grow_nl_kernels.hpp): kernels of their own, so
// that their registers (a 48-entry sorting network, 24 weights) and LDS (the window's dual values) cost the TV-l2 builds
// nothing.
template <int FN>
__global__ __launch_bounds__(kPatchThreads) void grow_kernel_nl(const GrowDir *__restrict__ dirs, int mode, int iteration, int wr,
                                                              int max_iter_patch, int budget, int ntop)
{
    __shared__ GrowShared sm;
    __shared__ NlShared ns;
    grow_body<FN>(sm, &ns, dirs, mode, iteration, wr, max_iter_patch, budget, ntop);
}

// insert_potential_candidates (src/local_faldoi.cpp:813-846) + prepare_data_for_growing (:860-870) for one growing per
// workgroup: every pixel with a finite flow is pushed, in row-major order, with its stored energy into the (empty)
// queue -- N successive push_heap calls, serial by nature, done by the first wavefront with its cooperative
// sift-up (64 pixels are fetched at a time, then visited in order) -- and the per-pixel state is reset.
__global__ __launch_bounds__(kPatchThreads) void grow_reseed_kernel(const GrowDir *__restrict__ dirs)
{
    const GrowDir D = dirs[blockIdx.x];
    const int t = threadIdx.x, n = D.nx * D.ny;
    if (t < 64) {
        int size = 0;
        for (int p0 = 0; p0 < n; p0 += 64) {
            const int p = p0 + t < n ? p0 + t : n - 1;
            const float u = D.out[p], v = D.out[D.pe + p], e = D.ene[p];
            const unsigned long long fin = __builtin_amdgcn_ballot_w64(p0 + t < n && __builtin_isfinite(u) && __builtin_isfinite(v));
            for (unsigned long long rest = fin; rest != 0ull; rest &= rest - 1) {
                const int q = __builtin_ctzll(rest), g = p0 + q;
                if (size < D.heap_cap) heap_push_wave(heap_global_only(D.heap), size, g % D.nx, g / D.nx, lane_read(u, q), lane_read(v, q), lane_read(e, q));
                else if (t == 0) D.ctl->overflow = 1;
            }
        }
        if (t == 0) {
            D.ctl->heap_size = size;
            D.ctl->done = 0;
        }
    }
    __syncthreads();
    for (int p = t; p < n; p += kPatchThreads) {
        D.fixed[p] = 0;
        D.ene[p] = __builtin_inff();
        D.out[p] = __builtin_nanf("");
        D.out[D.pe + p] = __builtin_nanf("");
    }
}

// Test hook: replays push (1) / pop (0) operations on the device queue with one wavefront.
__global__ __launch_bounds__(kPatchThreads) void heap_replay_kernel(Candidate *heap, int nops, const int *__restrict__ ops,
                                                                  const float *__restrict__ sims, int *__restrict__ popped,
                                                                  int *__restrict__ npopped, int ntop)
{
    // as in grow_body: two wavefronts, the first one runs the queue (its top `ntop` elements in LDS), its counters live in
    // vector registers
    extern __shared__ __attribute__((aligned(16))) unsigned char replay_heap_top[];
    const HeapRef h{ (heap_ptr)heap, (heap_lds_ptr)(reinterpret_cast<cand4 *>(replay_heap_top)), ntop };
#ifdef FALDOI_HEAP_STAMPS
    if (threadIdx.x < 16) heap_stamp_acc()[threadIdx.x] = 0ull;
    __syncthreads();
#endif
    if (threadIdx.x < 64) {
        int size = npopped[0], np = npopped[0];              // both 0 (the compiler cannot know)
        long long cyc_push = 0, cyc_pop = 0, n_push = 0, n_pop = 0;      // shader-clock cycles per kind (tools/heap_cycles.py)
        for (int k = 0; k < nops; ++k) {
            const long long c0 = clock64();
            if (ops[k]) {
                heap_push_wave(h, size, k & 0xffff, (k * 7) & 0x1ff, sims[k] * 3.0f + 1.0f, -sims[k], sims[k]);
                cyc_push += clock64() - c0;
                ++n_push;
            } else if (size == 0) {
                if (threadIdx.x == 0) popped[np] = -1;
                ++np;
            } else {
                const cand4 c = h.ld(0);
                heap_remove_top_wave(h, size);
                const unsigned cw = __float_as_uint(c.w);
                const bool intact = c.y == c.x * 3.0f + 1.0f && c.z == -c.x && (cw >> 16) == (((cw & 0xffffu) * 7) & 0x1ff);
                if (threadIdx.x == 0) popped[np] = intact ? (int)(cw & 0xffffu) : -2;     // payload travels with its key
                ++np;
                if (k >= nops / 2) {                         // (the second half of the replay: the queue has its size by then)
                    cyc_pop += clock64() - c0;
                    ++n_pop;
                }
            }
        }
        if (threadIdx.x == 0) {
            *npopped = np;
            long long *cyc = reinterpret_cast<long long *>(npopped + 2);
            cyc[0] = cyc_push; cyc[1] = n_push; cyc[2] = cyc_pop; cyc[3] = n_pop;
#ifdef FALDOI_HEAP_STAMPS
            for (int k = 0; k < 16; ++k) g_heap_stamp[k] = heap_stamp_acc()[k];
#endif
        }
    }
}

// Reset of insert_initial_seeds :755-761 / insert_potential_candidates :839-845 / prepare_data_for_growing.
__global__ void grow_reset_kernel(int *__restrict__ fixed, float *__restrict__ ene, float *__restrict__ out, size_t pe, int n)
{
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        fixed[p] = 0;
        ene[p] = __builtin_inff();
        out[p] = __builtin_nanf("");
        out[pe + p] = __builtin_nanf("");
    }
}

// Second half of insert_initial_seeds (:785-795): the seeds become fixed points with energy 0.
__global__ void grow_fix_seeds_kernel(const int *__restrict__ seeds, int nseeds, const float *__restrict__ seed_flow,
                                      int *__restrict__ fixed, float *__restrict__ ene, float *__restrict__ out, size_t pe)
{
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nseeds; s += gridDim.x * blockDim.x) {
        const int g = seeds[s];
        out[g] = seed_flow[g];
        out[pe + g] = seed_flow[pe + g];
        fixed[g] = 1;
        ene[g] = 0.0f;
    }
}

// fb_consistency_check (:167-190): trust = |u(x) + v(x + u(x))| <= epsilon, then
// delete_not_trustable_candidates (:283-311) on the same pixel.  hypotf is evaluated the way the C
// library of the reference build does for finite arguments: sqrt(x*x + y*y) in double, rounded to float.
__global__ void grow_prune_kernel(const float *__restrict__ in0, const float *__restrict__ in1, size_t pe, int nx, int ny,
                                  float epsilon, int *__restrict__ trust, unsigned *__restrict__ kept)
{
    const int n = nx * ny;
    unsigned mine = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const int y = p / nx, x = p - y * nx;
        const float a = in0[p], b = in0[pe + p];
        const BicubicTaps tp = bicubic_taps((float)x + a, (float)y + b, nx, ny);
        const float u1w = tp.clamped ? 0.0f : bicubic_eval_rows(in1, tp, nx);          // border_out = true
        const float u2w = tp.clamped ? 0.0f : bicubic_eval_rows(in1 + pe, tp, nx);
        const float sx = a + u1w, sy = b + u2w;
        float tolerance;
        if (__builtin_isinf(sx) || __builtin_isinf(sy)) tolerance = __builtin_inff();
        else tolerance = (float)__builtin_sqrt((double)sx * (double)sx + (double)sy * (double)sy);
        const int ok = (tolerance > epsilon) ? 0 : 1;
        trust[p] = ok;
        mine += ok;
    }
    if (mine) atomicAdd(kept, mine);
}

__global__ void grow_delete_kernel(const int *__restrict__ trust, float *__restrict__ out, float *__restrict__ u,
                                   float *__restrict__ ene, size_t pe, int n)
{
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x)
        if (trust[p] == 0) {
            out[p] = __builtin_nanf("");
            out[pe + p] = __builtin_nanf("");
            u[p] = __builtin_nanf("");
            u[pe + p] = __builtin_nanf("");
            ene[p] = __builtin_inff();
        }
}

}  // namespace faldoi
