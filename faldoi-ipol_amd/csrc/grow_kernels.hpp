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
    int pad;                          // set to 1 if a push did not fit (cannot happen: heap_cap is the worst case)
    long long pops, solves, fills;    // statistics
};

struct GrowDir {                      // per growing: device pointers (array in HBM, one entry per workgroup)
    float *planes;                    // [L_COUNT][pe]: source frame, target frame + gradient, flow, duals, divergences
    int *fixed;                       // OpticalFlowData::fixed_points
    int *trust;                       // OpticalFlowData::trust_points
    float *ene;                       // ene_Go / ene_Ba
    float *out;                       // oft0 / oft1: [2][pe]
    const float *sal;                 // saliency or nullptr (= 1)
    Candidate *heap;
    GrowCtl *ctl;
    const int *seeds;                 // pixel indices of the finite seeds, row-major order
    const float *seed_flow;           // [2][pe]
    int nseeds;
    int heap_cap;
};

struct GrowShared {
    float in[2][kElapPyr];            // harmonic fill: data pyramid (NaN = hole), per channel
    float y[2][kElapPyr];             // harmonic fill: solution pyramid
    float a[kPatchThreads], b[kPatchThreads];     // window exchange of ubar / u
    float x[4][kPatchThreads];        // duals of the current iteration
    float t[kPatchThreads];           // energy terms
    int err[2];                       // max |du|^2 (float bits), double-buffered
    int fmax[2];                      // harmonic fill: max update per channel (float bits)
    int cur[4];                       // {have, i, j, pixel}
    int flag;
};

// ---------------------------------------------------------------------------------------------
// interpolate_poisson on the window, all 128 threads.  Wave c fills flow channel c.  The sweeps of
// src/elap_recsep.c:50-69 update in row-major order and each update sees its left and upper
// neighbours already updated: pixels of one anti-diagonal are independent, so a sweep is run
// diagonal by diagonal with the lanes of the wave on the pixels of the diagonal -- same values,
// same order of dependence.  Returns this thread's filled (u1, u2) (threads outside the window: 0).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void harmonic_fill_window(GrowShared &sm, const float *out, size_t pe, const int *fixed, int nx,
                                                     int ii, int ij, int pw, int ph, float &u1, float &u2)
{
    const int t = threadIdx.x, c = t >> 6, lane = t & 63;
    int ws[kElapLevels], hs[kElapLevels], off[kElapLevels];
    ws[0] = pw; hs[0] = ph; off[0] = 0;
#pragma unroll
    for (int s = 1; s < kElapLevels; ++s) {
        ws[s] = (ws[s - 1] + 1) / 2;                               // ceil(w / 2.0), src/elap_recsep.c:195-196
        hs[s] = (hs[s - 1] + 1) / 2;
        off[s] = off[s - 1] + ws[s - 1] * hs[s - 1];
    }
    float *in = sm.in[c], *y = sm.y[c];
    const float *src = out + (size_t)c * pe;
    for (int q = lane; q < pw * ph; q += 64) {                      // copy_fixed_coordinates + :341-355
        const int xy = (ij + q / pw) * nx + ii + q % pw;
        in[q] = (fixed[xy] == 1) ? src[xy] : __builtin_nanf("");
    }
    __syncthreads();
#pragma unroll
    for (int s = 1; s < kElapLevels; ++s) {                         // zoom_out_by_factor_two :160-175
        const int iw = ws[s - 1], ih = hs[s - 1], ow = ws[s];
        if (lane < ow * hs[s]) {
            const int i = lane % ow, j = lane / ow;
            const float *x = in + off[s - 1];
            const float a4[4] = { px_clamp(x, iw, ih, 2 * i, 2 * j), px_clamp(x, iw, ih, 2 * i + 1, 2 * j),
                                  px_clamp(x, iw, ih, 2 * i, 2 * j + 1), px_clamp(x, iw, ih, 2 * i + 1, 2 * j + 1) };
            float m = 0;
            int cx = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (__builtin_isfinite(a4[q])) { m += a4[q]; cx += 1; }
            in[off[s] + lane] = cx ? m / cx : __builtin_nanf("");
        }
        __syncthreads();
    }
#pragma unroll
    for (int s = kElapLevels - 1; s >= 0; --s) {
        const int w = ws[s], h = hs[s];
        const float *x = in + off[s];
        float *ys = y + off[s];
        for (int q = lane; q < w * h; q += 64) {                    // :113-114 with the initialisation of :204-222
            float v = x[q];
            if (!__builtin_isfinite(v)) {
                v = 0.0f;                                           // coarsest level starts from zero
                if (s < kElapLevels - 1)                            // round((i - 0.5) / 2) == i / 2 for i >= 0
                    v = px_clamp(y + off[s + 1], ws[s + 1], hs[s + 1], (q % w) >> 1, (q / w) >> 1);
            }
            ys[q] = v;
        }
        if (lane == 0) sm.fmax[c] = 0;
        __syncthreads();
        bool stop = false;
        for (int it = 0; it < 3; ++it) {                            // niter = 3, timestep 0.4
            float mx = 0.0f;
            for (int d = 0; d < w + h - 1; ++d) {
                const int i = lane, j = d - lane;
                if (!stop && i < w && j >= 0 && j < h && __builtin_isnan(x[j * w + i])) {
                    const int idx = j * w + i;
                    const float lap = -4 * px_clamp(ys, w, h, i, j) + px_clamp(ys, w, h, i + 1, j) + px_clamp(ys, w, h, i, j + 1)
                                      + px_clamp(ys, w, h, i - 1, j) + px_clamp(ys, w, h, i, j - 1);
                    const float nw = ys[idx] + 0.4f * lap;
                    const float update = __builtin_fabsf(ys[idx] - nw);
                    if (update > mx) mx = update;
                    ys[idx] = nw;
                }
                __syncthreads();
            }
            if (!stop && mx > 0.0f) atomicMax(&sm.fmax[c], __float_as_int(mx));
            __syncthreads();
            if ((double)__int_as_float(sm.fmax[c]) < 1e-10) stop = true;         // :66-67
            __syncthreads();
            if (lane == 0) sm.fmax[c] = 0;
        }
        __syncthreads();
    }
    u1 = 0.0f;
    u2 = 0.0f;
    if (t < pw * ph) {
        u1 = sm.y[0][t];
        u2 = sm.y[1][t];
    }
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
__device__ __forceinline__ float window_solve(GrowShared &sm, float *base, size_t pe, int nx, int ny, int ii, int ij, int ei,
                                              int ej, int max_iter_patch, float &u1, float &u2)
{
    const int pw = ei - ii, ph = ej - ij, t = threadIdx.x;
    const bool act = t < pw * ph;
    const int k = ii + (act ? t % pw : 0), l = ij + (act ? t / pw : 0);
    const int p = l * nx + k;
    const float lambda = 40.0f, theta = 0.3f, tau = 0.125f, tol_OF = 0.01f;        // src/energy_model.cpp:704-709
    const float l_t = lambda * theta;
    const float *I0 = base + L_I0 * pe, *I1 = base + L_I1 * pe, *I1x = base + L_I1X * pe, *I1y = base + L_I1Y * pe;
    float *D1 = base + L_DIV1 * pe, *D2 = base + L_DIV2 * pe;

    // which divergence statement of src/utils.cpp:76-108 writes this thread's own entry
    const bool k_in = k > ii && k < ei - 1, l_in = l > ij && l < ej - 1;
    const bool first_row = l == ij, last_row = l == ej - 1, first_col = k == ii, last_col = k == ei - 1;
    const bool corner = (first_row || last_row) && (first_col || last_col);
    const bool fresh = act && ((k_in && l_in) || (k_in && last_row) || corner || (k_in && first_row && ij == 0) ||
                               (l_in && first_col && ii == 0) || (l_in && last_col && ei == nx));

    float x11 = 0, x12 = 0, x21 = 0, x22 = 0, rc = 0, wx = 0, wy = 0, gr = 0, v1 = 0, v2 = 0, ub1 = 0, ub2 = 0, d1 = 0, d2 = 0,
          i0 = 0;
    if (act) {
        d1 = D1[p];
        d2 = D2[p];
        i0 = I0[p];
        const float uu = (float)k + u1, vv = (float)l + u2;                       // src/bicubic_interpolation.c:288-289
        const BicubicTaps tp = bicubic_taps(uu, vv, nx, ny);                      // border_out = false
        const float w0 = bicubic_eval(I1, tp, nx);
        wx = bicubic_eval(I1x, tp, nx);
        wy = bicubic_eval(I1y, tp, nx);
        gr = grad_sq(wx, wy);
        rc = rho_const(w0, wx, wy, u1, u2, i0);
        ub1 = u1;
        ub2 = u2;
    }
    int it = 0;
    float err_D = __builtin_inff();
    while (err_D > tol_OF * tol_OF && it < max_iter_patch) {                      // src/tvl2_model.cpp:359
        ++it;
        if (act) {                                                                // :364-391
            const float rho = rc + (wx * u1 + wy * u2);
            float e1, e2;
            if (rho < -l_t * gr) { e1 = l_t * wx; e2 = l_t * wy; }
            else if (rho > l_t * gr) { e1 = -l_t * wx; e2 = -l_t * wy; }
            else if (gr < grad_zero_bound()) { e1 = 0.0f; e2 = 0.0f; }
            else { const float fi = -rho / gr; e1 = fi * wx; e2 = fi * wy; }
            v1 = u1 + e1;
            v2 = u2 + e2;
        }
        sm.a[t] = ub1;
        sm.b[t] = ub2;
        if (t == 0) sm.err[it & 1] = 0;
        __syncthreads();
        if (act) {                                                                // forward_gradient_patch + tvl2coupled_getD
            const float u1x = !last_col ? sm.a[t + 1] - ub1 : 0.0f, u1y = !last_row ? sm.a[t + pw] - ub1 : 0.0f;
            const float u2x = !last_col ? sm.b[t + 1] - ub2 : 0.0f, u2y = !last_row ? sm.b[t + pw] - ub2 : 0.0f;
            const float g11 = x11 * x11, g12 = x12 * x12, g21 = x21 * x21, g22 = x22 * x22;
            float n = __builtin_sqrtf(g11 + g12 + g21 + g22);
            n = (1.0f > n) ? 1.0f : n;
            x11 = (x11 + tau * u1x) / n;
            x12 = (x12 + tau * u1y) / n;
            x21 = (x21 + tau * u2x) / n;
            x22 = (x22 + tau * u2y) / n;
        }
        sm.x[0][t] = x11; sm.x[1][t] = x12; sm.x[2][t] = x21; sm.x[3][t] = x22;
        __syncthreads();
        float uN = 0.0f;
        if (act) {
            if (fresh) {                                                          // divergence_patch, src/utils.cpp:63-113
                const float l11 = first_col ? 0.0f : sm.x[0][t - 1], l21 = first_col ? 0.0f : sm.x[2][t - 1];
                const float t12 = first_row ? 0.0f : sm.x[1][t - pw], t22 = first_row ? 0.0f : sm.x[3][t - pw];
                if (corner) {                                                     // :104-108
                    if (first_col && first_row) { d1 = x11 + x12; d2 = x21 + x22; }
                    else if (last_col && first_row) { d1 = -l11 + x12; d2 = -l21 + x22; }
                    else if (first_col && last_row) { d1 = x11 - t12; d2 = x21 - t22; }
                    else { d1 = -l11 - t12; d2 = -l21 - t22; }
                } else if (k_in && l_in) {                                        // :76-86
                    d1 = (x11 - l11) + (x12 - t12);
                    d2 = (x21 - l21) + (x22 - t22);
                } else if (k_in && last_row) {                                    // :91-92
                    d1 = x11 - l11 - t12;
                    d2 = x21 - l21 - t22;
                } else if (k_in) {                                                // :90, image row 0
                    d1 = x11 - l11 + x12;
                    d2 = x21 - l21 + x22;
                } else if (first_col) {                                           // :98-99, image column 0
                    d1 = x11 + x12 - t12;
                    d2 = x21 + x22 - t22;
                } else {                                                          // :100-102, image column nx-1
                    d1 = -l11 + x12 - t12;
                    d2 = -l21 + x22 - t22;
                }
            }
            const float u1k = u1, u2k = u2;                                       // tvl2coupled_getP :122-166
            u1 = u1k - tau * (-d1 + (u1k - v1) / theta);
            u2 = u2k - tau * (-d2 + (u2k - v2) / theta);
            uN = (u1 - u1k) * (u1 - u1k) + (u2 - u2k) * (u2 - u2k);
            ub1 = 2.0f * u1 - u1k;                                                // :420-427
            ub2 = 2.0f * u2 - u2k;
            if (uN > 0.0f) atomicMax(&sm.err[it & 1], __float_as_int(uN));        // NaN never wins, as in the reference
        }
        __syncthreads();
        err_D = __int_as_float(sm.err[it & 1]);
    }

    // ---- eval_tvl2coupled (src/tvl2_model.cpp:174-243) ----
    sm.a[t] = u1;
    sm.b[t] = u2;
    __syncthreads();
    float term = 0.0f;
    if (act) {
        base[L_U1 * pe + p] = u1;
        base[L_U2 * pe + p] = u2;
        D1[p] = d1;
        D2[p] = d2;
        const float u1x = !last_col ? sm.a[t + 1] - u1 : 0.0f, u1y = !last_row ? sm.a[t + pw] - u1 : 0.0f;
        const float u2x = !last_col ? sm.b[t + 1] - u2 : 0.0f, u2y = !last_row ? sm.b[t + pw] - u2 : 0.0f;
        const BicubicTaps tp = bicubic_taps((float)k + u1, (float)l + u2, nx, ny);
        const float iw = bicubic_eval(I1, tp, nx);
        const float dt = lambda * __builtin_fabsf(iw - i0);
        const float dc = (1 / (2 * theta)) * ((u1 - v1) * (u1 - v1) + (u2 - v2) * (u2 - v2));
        const float g = __builtin_sqrtf(u1x * u1x + u1y * u1y + u2x * u2x + u2y * u2y);
        term = dc + dt + g;
    }
    sm.t[t] = term;
    __syncthreads();
    float ener = 0.0f;
    if (t == 0) {                                        // the reference accumulates in row-major order, in fp32
        const int m = pw * ph;
        for (int q = 0; q < m; ++q) ener += sm.t[q];
        ener = (float)((double)ener / (m * 1.0));
    }
    return ener;
}

// add_neighbors (src/local_faldoi.cpp:679-727) for the pixel (i, j), all 128 threads.
__device__ __forceinline__ void add_neighbors_dev(GrowShared &sm, const GrowDir &D, size_t pe, int nx, int ny, int i, int j,
                                                  int iteration, int wr, int max_iter_patch, int &heap_size,
                                                  long long &fills)
{
    const int t = threadIdx.x;
    const int ii = (i - wr < 0) ? 0 : i - wr, ij = (j - wr < 0) ? 0 : j - wr;             // get_index_patch
    const int ei = (i + 1 + wr > nx) ? nx : i + 1 + wr, ej = (j + 1 + wr > ny) ? ny : j + 1 + wr;
    const int pw = ei - ii, ph = ej - ij;
    const bool act = t < pw * ph;
    const int p = (ij + (act ? t / pw : 0)) * nx + ii + (act ? t % pw : 0);
    bool fill = iteration == 0;
    if (!fill) fill = __syncthreads_or(act && D.trust[p] == 0);                   // check_trustable_patch :639-656
    float u1 = 0.0f, u2 = 0.0f;
    if (fill) {
        harmonic_fill_window(sm, D.out, pe, D.fixed, nx, ii, ij, pw, ph, u1, u2);
        if (t == 0) ++fills;
    } else if (act) {
        u1 = D.planes[L_U1 * pe + p];
        u2 = D.planes[L_U2 * pe + p];
    }
    const float ener = window_solve(sm, D.planes, pe, nx, ny, ii, ij, ei, ej, max_iter_patch, u1, u2);
    if (t == 0) {                                                                 // insert_candidates :497-537
        const int nbx[4] = { 0, 0, 1, -1 }, nby[4] = { 1, -1, 0, 0 };
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int px = i + nbx[q], py = j + nby[q];
            if (px < 0 || px >= nx || py < 0 || py >= ny) continue;
            const int g = py * nx + px;
            const float new_ener = ener * (D.sal ? D.sal[g] : 1.0f);
            if (!D.fixed[g] && new_ener < D.ene[g]) {
                D.ene[g] = ener;
                const int wq = (py - ij) * pw + (px - ii);                        // the neighbour lies in the window
                if (heap_size < D.heap_cap) heap_push(D.heap, heap_size, make_candidate(px, py, sm.a[wq], sm.b[wq], new_ener));
                else D.ctl->pad = 1;
            }
        }
        const int g = j * nx + i, wq = (j - ij) * pw + (i - ii);                  // :718-726
        if (D.ene[g] > ener) {
            D.out[g] = sm.a[wq];
            D.out[pe + g] = sm.b[wq];
            D.ene[g] = ener;
        }
    }
    __syncthreads();
}

// mode 0: the seed pass of insert_initial_seeds (:764-783); mode 1: local_growing (:899-937).
__global__ __launch_bounds__(kPatchThreads) void grow_kernel(const GrowDir *__restrict__ dirs, size_t pe, int nx, int ny,
                                                           int mode, int iteration, int wr, int max_iter_patch, int budget)
{
    __shared__ GrowShared sm;
    const GrowDir D = dirs[blockIdx.x];
    const int t = threadIdx.x;
    int heap_size = 0, seed_pos = 0;
    long long pops = 0, solves = 0, fills = 0;
    if (t == 0) {
        heap_size = D.ctl->heap_size;
        seed_pos = D.ctl->seed_pos;
    }
    int done = 0;
    for (int step = 0; step < budget; ++step) {
        if (t == 0) {
            int have = 0, g = 0;
            if (mode == 0) {
                if (seed_pos < D.nseeds) {
                    g = D.seeds[seed_pos++];
                    D.out[g] = D.seed_flow[g];
                    D.out[pe + g] = D.seed_flow[pe + g];
                    D.fixed[g] = 1;
                    have = 1;
                }
            } else {
                while (heap_size > 0) {
                    const Candidate c = heap_pop(D.heap, heap_size);
                    ++pops;
                    const int ci = (int)(c.ij & 0xffffu), cj = (int)(c.ij >> 16);
                    g = cj * nx + ci;
                    if (D.fixed[g]) continue;
                    D.fixed[g] = 1;                                               // :926-932
                    D.out[g] = c.u;
                    D.out[pe + g] = c.v;
                    D.ene[g] = c.sim;
                    have = 1;
                    break;
                }
            }
            sm.cur[0] = have;
            sm.cur[1] = g % nx;
            sm.cur[2] = g / nx;
        }
        __syncthreads();
        if (!sm.cur[0]) { done = 1; break; }
        const int i = sm.cur[1], j = sm.cur[2];
        add_neighbors_dev(sm, D, pe, nx, ny, i, j, iteration, mode == 0 ? 1 : wr, max_iter_patch, heap_size, fills);
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
    if (t == 0) {
        if (!done) done = (mode == 0) ? (seed_pos >= D.nseeds) : (heap_size == 0);
        D.ctl->heap_size = heap_size;
        D.ctl->seed_pos = seed_pos;
        D.ctl->done = done;
        D.ctl->pops += pops;
        D.ctl->solves += solves;
        D.ctl->fills += fills;
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
        const float u1w = tp.clamped ? 0.0f : bicubic_eval(in1, tp, nx);          // border_out = true
        const float u2w = tp.clamped ? 0.0f : bicubic_eval(in1 + pe, tp, nx);
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
