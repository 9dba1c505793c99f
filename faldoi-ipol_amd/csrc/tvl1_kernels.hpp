// tvl1_kernels.hpp -- HIP kernels of the TV-L1 global minimisation (gfx950 / MI355X).
//
// Design: the reference runs seven full-image passes per inner iteration
// (src/global_faldoi.cpp:690-783, 208 B/px/iteration of traffic).  Here one pointwise launch per
// outer warp builds the warp constants, and ONE launch advances a
// tile by up to FALDOI_MAX_CHUNK inner iterations with the whole per-pixel state held in
// registers (temporal blocking): a workgroup loads its tile plus a halo as wide as the number
// of iterations, iterates without touching HBM, and writes back only the tile interior, whose
// values are exactly what the full-image sweeps would have produced.  The halo absorbs the
// one-pixel-per-iteration growth of the stencil's dependency cone, so workgroups never
// communicate inside a launch.
//
// Lane layout: tile width = 64 = one wavefront; lane l owns column l; wave w owns rows
// [w*R, w*R+R) of that column in registers.  Horizontal neighbours come from the adjacent
// lane (wavefront shuffle), vertical neighbours from the thread's own registers, except
// across wave boundaries where one row per wave is exchanged through LDS.
#pragma once
#include "faldoi_device.hpp"
#include <type_traits>

#pragma clang fp contract(off)

namespace faldoi {

constexpr int kTileW = 64;          // one wavefront wide
constexpr int kMaxChunk = 16;       // == FALDOI_MAX_CHUNK

// Plane indices inside one pair's slab (each plane = plane_elems floats).
enum Plane {
    P_I0 = 0, P_I1, P_I1X, P_I1Y,            // inputs + centred gradient of I1 (per solve)
    P_RC, P_IWX, P_IWY,                      // per-warp constants: rho_c, I1wx, I1wy
    P_STATE0,                                // state buffer 0: u1 u2 ub1 ub2 xi11 xi12 xi21 xi22
    P_STATE1 = P_STATE0 + 8,                 // state buffer 1 (ping-pong)
    P_INIT = P_STATE1 + 8,                   // uploaded u1 u2 xi11..xi22 (for batch_reset)
    P_COUNT = P_INIT + 6
};

struct TvConst {
    double rtheta;                 // 1 / (double)theta, for the exact fp64-assisted quotient
    int once;                      // host side only: launch the instantiations with the one-product quotients (ONCE)
    float l_t, theta, tau, tol2;
    int max_inner, tol_stop, warps, chunk;
};

// Per-pair control block, owned by the device between launches in "dynamic" (tolerance stop)
// mode; in static mode the host passes the same fields as kernel arguments.
struct PairCtl {
    int warp;        // current outer warp
    int n;           // inner iterations completed in this warp
    int cur;         // which state buffer holds the current state
    int first;       // next chunk starts a warp (do the bicubic warp, ubar = u)
    int iters;       // inner iterations the next chunk launch must run (0 = nothing to do)
    int redo;        // next chunk is the exact-length re-run that ends the warp
    int done;        // all warps finished
    int unsafe;      // raised by the kernels that compute warp constants: a |grad I|^2 that does not admit the one-product quotient
};

// ---------------------------------------------------------------------------------------------
// Small stand-alone kernels (per-solve setup and the C-ABI primitives).
// ---------------------------------------------------------------------------------------------

// src/utils.cpp:367 centered_gradient.  Grid-stride over pixels of `npairs` planes.
__global__ void centered_gradient_kernel(const float *__restrict__ in, float *__restrict__ dx,
                                         float *__restrict__ dy, int nx, int ny,
                                         size_t in_stride, size_t out_stride, int npairs)
{
    const size_t n = (size_t)nx * ny;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n * npairs;
         i += (size_t)gridDim.x * blockDim.x) {
        const int pair = (int)(i / n);
        const int p = (int)(i - (size_t)pair * n);
        const int y = p / nx, x = p - y * nx;
        const float *I = in + pair * in_stride;
        const int xl = x > 0 ? x - 1 : x, xr = x < nx - 1 ? x + 1 : x;
        const int yu = y > 0 ? y - 1 : y, yd = y < ny - 1 ? y + 1 : y;
        dx[pair * out_stride + p] = centered_diff(I[y * nx + xr], I[y * nx + xl]);
        dy[pair * out_stride + p] = centered_diff(I[yd * nx + x], I[yu * nx + x]);
    }
}

// src/utils.cpp:285 forward_gradient.
__global__ void forward_gradient_kernel(const float *__restrict__ f, float *__restrict__ fx,
                                        float *__restrict__ fy, int nx, int ny)
{
    const int n = nx * ny;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const int y = p / nx, x = p - y * nx;
        const float s = f[p];
        const float r = (x < nx - 1) ? f[p + 1] : 0.0f;
        const float d = (y < ny - 1) ? f[p + nx] : 0.0f;
        fx[p] = fwd_diff(r, s, x == nx - 1);
        fy[p] = fwd_diff(d, s, y == ny - 1);
    }
}

// src/utils.cpp:239 divergence.
__global__ void divergence_kernel(const float *__restrict__ v1, const float *__restrict__ v2,
                                  float *__restrict__ dv, int nx, int ny)
{
    const int n = nx * ny;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const int y = p / nx, x = p - y * nx;
        const float al = (x > 0) ? v1[p - 1] : 0.0f;
        const float bu = (y > 0) ? v2[p - nx] : 0.0f;
        dv[p] = div_sample(v1[p], al, v2[p], bu, classify(x, y, nx, ny));
    }
}

// src/bicubic_interpolation.c:245 bicubic_interpolation_warp.
__global__ void bicubic_warp_kernel(const float *__restrict__ in, const float *__restrict__ u,
                                    const float *__restrict__ v, float *__restrict__ out,
                                    int nx, int ny, int border_out)
{
    const int n = nx * ny;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const int y = p / nx, x = p - y * nx;
        const float uu = (float)x + u[p];
        const float vv = (float)y + v[p];
        const BicubicTaps t = bicubic_taps(uu, vv, nx, ny);
        out[p] = (t.clamped && border_out) ? 0.0f : (t.clamped ? bicubic_eval(in, t, nx) : bicubic_eval_rows(in, t, nx));
    }
}

// Self-test of the fp64-assisted quotient (faldoi_device.hpp: recip_f64/div_by) against the
// hardware IEEE fp32 divide.  Sample i draws a numerator from a counter-based hash over ALL finite
// bit patterns (every exponent incl. subnormals, both signs; every 16th sample is a special value)
// and divides it (a) by max(1,|x|) for a hashed x, the dual step's divisor, and (b) by a hashed
// theta in (2^-20, 4), the primal step's divisor.  Counts bitwise mismatches.  mismatch[0] total, [1] dual site, [2] subnormal
// results, [3] huge divisors, [4] dump cursor, [5] one-product form, [6] divisors that form refuses, [7] ... of which it would get wrong.
__device__ __forceinline__ unsigned mix32(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void selftest_division_kernel(unsigned seed, unsigned long long count, unsigned long long *mismatch,
                                         float *dump, int dump_cap)
{
    unsigned long long bad = 0, bad_dual = 0, bad_sub = 0, bad_big = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned h0 = mix32((unsigned)i ^ seed), h1 = mix32(h0 + 0x9e3779b9u + (unsigned)(i >> 32));
        const unsigned h2 = mix32(h1 ^ 0x85ebca6bu);
        unsigned ab = h0;
        if ((i & 15) == 0) {                                  // specials: zeros, subnormals, tie makers
            const unsigned sel = (h1 >> 8) & 7;
            ab = sel == 0 ? 0u : sel == 1 ? 0x80000000u : sel == 2 ? (h0 & 0x807fffffu)       // +-subnormal
               : sel == 3 ? ((h0 & 0x80000007u) | 0x00000001u)                                 // tiny subnormal
               : sel == 4 ? ((h0 & 0x80ffffffu) | 0x00800000u)                                 // smallest normals
               : sel == 5 ? ((h0 & 0x807fffffu) | 0x7f000000u)                                 // huge
               : sel == 6 ? (h0 & 0xff800000u) : h0;                                           // powers of two
        }
        if ((ab & 0x7f800000u) == 0x7f800000u) ab &= 0xbfffffffu;     // keep it finite
        const float a = __uint_as_float(ab);
        // (a) divisor of the dual step
        unsigned xb = h1;
        if ((xb & 0x7f800000u) == 0x7f800000u) xb &= 0xbfffffffu;
        if ((i & 3) == 1) xb = (xb & 0x007fffffu) | 0x3f800000u | ((h2 & 3u) << 23);   // |x| in [1,16): the common case
        float n = fabsf(__uint_as_float(xb));
        n = (1.0f > n) ? 1.0f : n;
        const float q_ref = a / n;
        const float q_new = div_by(a, n, recip_f64(n));
        const bool m1 = (__float_as_uint(q_ref) != __float_as_uint(q_new));
        bad += m1;
        bad_dual += m1;
        bad_sub += m1 && (fabsf(q_ref) < 1.17549435e-38f);
        bad_big += m1 && (n > 1e30f);
        if (m1 && dump) {                                     // first few offenders, for diagnosis
            const unsigned long long slot = atomicAdd(mismatch + 4, 1ull);
            if (slot < (unsigned long long)dump_cap) {
                dump[slot * 4 + 0] = a; dump[slot * 4 + 1] = n; dump[slot * 4 + 2] = q_ref; dump[slot * 4 + 3] = q_new;
            }
        }
        // (b) divisor of the primal step
        const float theta = __uint_as_float(0x35800000u + (h2 % 0x0b000000u));          // [2^-20, 4)
        const float t_ref = a / theta;
        const float t_new = div_by(a, theta, 1.0 / (double)theta);
        const bool m2 = (__float_as_uint(t_ref) != __float_as_uint(t_new));
        bad += m2;
        bad_sub += m2 && (fabsf(t_ref) < 1.17549435e-38f);
    }
    // (c) the one-product form (div_once): theta as above; a divisor as |grad I|^2 can be -- any magnitude in [2^-30, 2^30), every
    // fourth a small integer (the even ones >= 6 must be refused by div_once_ok, and are counted); numerators: the hashed one and a
    // tie maker d * (2k+1) * 2^-150 rounded to a float (an exact tie whenever that product is a float).  recip_f64_full must be within
    // one fp64 ulp of the correctly rounded reciprocal.
    unsigned long long bad_once = 0, refused = 0, forced = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned h0 = mix32((unsigned)i ^ seed ^ 0x5bd1e995u), h1 = mix32(h0 + 0x9e3779b9u + (unsigned)(i >> 32));
        const unsigned h2 = mix32(h1 ^ 0xc2b2ae35u);
        unsigned ab = h0;
        if ((i & 15) == 0) ab = (h0 & 0x80ffffffu);                    // subnormals and the smallest normals
        if ((ab & 0x7f800000u) == 0x7f800000u) ab &= 0xbfffffffu;
        float d = __uint_as_float(((97u + (h2 >> 8) % 60u) << 23) | (h1 & 0x007fffffu));
        if ((i & 3) == 2) d = (float)(1u + (h1 % 4096u)) * (float)(1u << ((h2 >> 4) & 3u));
        if ((i & 63) == 5) d = 0.3f;
        const double rd = recip_f64_full(d);
        const double rn = 1.0 / (double)d;
        long long ulps = (long long)__double_as_longlong(rd) - (long long)__double_as_longlong(rn);
        bad_once += (ulps > 1 || ulps < -1);
        const float tie = (float)((double)d * (double)(2u * (h0 % 1000u) + 1u) * 0x1p-150);
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const float a = w ? (((h2 >> 1) & 1u) ? -tie : tie) : __uint_as_float(ab);
            const bool differs = __float_as_uint(a / d) != __float_as_uint(div_once(a, rd));
            if (div_once_ok(d)) bad_once += differs;
            else { refused += 1; forced += differs; }
        }
    }
    // ... and every numerator for the default theta (NaN numerators: any NaN is accepted)
    {
        const float theta = 0.3f;
        const double rtheta = 1.0 / (double)theta;
        for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
             i += (unsigned long long)gridDim.x * blockDim.x) {
            const float a = __uint_as_float((unsigned)i);
            const float q_ref = a / theta, q_new = div_once(a, rtheta);
            bad_once += (__float_as_uint(q_ref) != __float_as_uint(q_new)) && !(q_ref != q_ref && q_new != q_new);
        }
    }
    bad += bad_once;
    if (refused) atomicAdd(mismatch + 6, refused);
    if (forced) atomicAdd(mismatch + 7, forced);
    if (bad_once) atomicAdd(mismatch + 5, bad_once);
    // exhaustive: sqrt(g*g) == |g| for every float the fast path of sqrt_of_square accepts
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32);
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float g = __uint_as_float((unsigned)i);
        if (sqrt_of_square_is_abs(g))
            bad += (__float_as_uint(__builtin_sqrtf(g * g)) != __float_as_uint(__builtin_fabsf(g)));
    }
    // exhaustive: sqrt_ge1 against sqrtf for every float in [1, FLT_MAX] (and +inf)
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
         i <= 0x7f800000ull - 0x3f800000ull; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(0x3f800000u + (unsigned)i);
        bad += (__float_as_uint(sqrt_ge1(x)) != __float_as_uint(__builtin_sqrtf(x)));
    }
    if (bad) {
        atomicAdd(mismatch, bad);
        atomicAdd(mismatch + 1, bad_dual);
        atomicAdd(mismatch + 2, bad_sub);
        atomicAdd(mismatch + 3, bad_big);
    }
}

// Copies the uploaded initial flow/duals into state buffer 0 and resets the control blocks.
__global__ void reset_state_kernel(float *__restrict__ base, size_t plane_elems, int npix, int npairs,
                                   PairCtl *__restrict__ ctl, int first_iters)
{
    const size_t total = (size_t)npix * npairs;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int pair = (int)(i / npix);
        const int p = (int)(i - (size_t)pair * npix);
        float *slab = base + (size_t)pair * P_COUNT * plane_elems;
        const float *ini = slab + (size_t)P_INIT * plane_elems;
        float *st = slab + (size_t)P_STATE0 * plane_elems;
        st[0 * plane_elems + p] = ini[0 * plane_elems + p];            // u1
        st[1 * plane_elems + p] = ini[1 * plane_elems + p];            // u2
#pragma unroll
        for (int k = 0; k < 4; ++k)                                    // xi11..xi22
            st[(4 + k) * plane_elems + p] = ini[(2 + k) * plane_elems + p];
    }
    if (blockIdx.x == 0 && threadIdx.x < npairs) {
        PairCtl c;
        c.warp = 0; c.n = 0; c.cur = 0; c.first = 1; c.iters = first_iters; c.redo = 0; c.done = 0; c.unsafe = 0;
        ctl[threadIdx.x] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// Per-warp kernel: bicubic warp of I1, I1x, I1y by the current flow (border_out = true), the
// per-warp constants rho_c / I1wx / I1wy  (src/global_faldoi.cpp:635-657).
// Pointwise in the flow, so it runs once per pixel (no halo).  In dynamic mode a pair takes part
// only when its control block says the next chunk starts a warp.
// ---------------------------------------------------------------------------------------------
template <int WPS>
__global__ __launch_bounds__(256, WPS) void tvl1_warp_kernel(float *__restrict__ base, size_t plane_elems, int nx, int ny,
                                                          int npairs, PairCtl *__restrict__ ctl, int static_mode,
                                                          int s_cur)
{
    const int npix = nx * ny;
    const int per_pair = (npix + 255) / 256;                   // blocks per pair
    const int pair = blockIdx.x / per_pair;
    if (pair >= npairs) return;
    int cur = s_cur;
    if (!static_mode) {
        const PairCtl c = ctl[pair];
        if (!c.first || c.iters <= 0) return;                   // uniform across the block
        cur = c.cur;
    }
    const int p = (blockIdx.x - pair * per_pair) * 256 + threadIdx.x;
    if (p >= npix) return;
    const int y = p / nx, x = p - y * nx;
    float *slab = base + (size_t)pair * P_COUNT * plane_elems;
    float *st = slab + (size_t)(cur ? P_STATE1 : P_STATE0) * plane_elems;
    const float *I0 = slab + (size_t)P_I0 * plane_elems;
    const float *I1 = slab + (size_t)P_I1 * plane_elems;
    const float *I1x = slab + (size_t)P_I1X * plane_elems;
    const float *I1y = slab + (size_t)P_I1Y * plane_elems;
    const float u1 = st[0 * plane_elems + p];
    const float u2 = st[1 * plane_elems + p];
    float i0v = I0[p];
    asm volatile("" : "+v"(i0v));                               // fetched with the flow, not in a round trip of its own at the end
    const float uu = (float)x + u1;                             // src/bicubic_interpolation.c:259-260
    const float vv = (float)y + u2;
    const BicubicTaps tp = bicubic_taps(uu, vv, nx, ny);
    float w0 = 0.0f, wx = 0.0f, wy = 0.0f;                      // border_out = true (:635-637)
    if (!tp.clamped) {
        w0 = bicubic_eval_rows(I1, tp, nx);
        wx = bicubic_eval_rows(I1x, tp, nx);
        wy = bicubic_eval_rows(I1y, tp, nx);
    }
    slab[(size_t)P_RC * plane_elems + p] = rho_const(w0, wx, wy, u1, u2, i0v);
    slab[(size_t)P_IWX * plane_elems + p] = wx;
    slab[(size_t)P_IWY * plane_elems + p] = wy;
    // a divisor the one-product quotient must not be used with (an even integer >= 6, faldoi_device.hpp): tell the host, which
    // repeats the solve with the general form if it ran the other one
    if (!div_once_ok(grad_sq(wx, wy))) ctl[pair].unsafe = 1;
    // ubar = u (:659-660) is not materialised: the first chunk of a warp takes ubar from u.
}

// ---------------------------------------------------------------------------------------------
// The fused iteration kernel: `iters` primal-dual iterations on a register-resident tile.
// ---------------------------------------------------------------------------------------------
//
// Grid: 8 * ceil(npairs*ntx*nty / 8) blocks of 64*NW threads.  Blocks are dealt to the 8 XCDs
// round-robin, so block b -> logical tile (b % 8) * per + b / 8 keeps spatially adjacent tiles
// (which share halo lines) on one XCD's L2.  Placement only affects speed.
//
// Tile (tx,ty) of a pair: origin (tx*Sx - halo, ty*Sy - halo), Sx = 64 - 2*halo,
// Sy = NW*R - 2*halo; it owns (stores) image pixels [tx*Sx,(tx+1)*Sx) x [ty*Sy,(ty+1)*Sy).
// After i iterations the values within distance i of a tile edge that is not an image edge are
// no longer the full-image values; with iters <= halo they never reach the owned region.
//
// err_out[(err_slot*ntiles_total + tile_global) * kMaxChunk + i] = max |du|^2 over the owned
// pixels after iteration i of this launch (plain stores, reduced by tvl1_ctl_kernel).
// BORDER = false is the specialisation for tiles whose 64 x TH footprint contains no pixel of
// the image frame (first/last row or column): every stencil uses the interior formula, no
// per-pixel class selects.  Chosen per workgroup (uniform branch) by the kernel below.
// The register-resident state of one thread: R rows of its column.
template <int R>
struct TileState {
    float u1[R], u2[R], ub1[R], ub2[R], x11[R], x12[R], x21[R], x22[R];   // flow, over-relaxed flow, duals
    float rc[R], ix[R], iy[R], gr[R];                                      // warp constants, |grad I|^2
    double rgr[R];                                                         // fp64 reciprocal of gr (recip_f64 / _full), constant over a warp
    bool own[R];                                                           // pixel stored by this tile
};

// Wave-uniform: every |grad I|^2 held by this wave admits the one-product quotient (faldoi_device.hpp: div_once_ok).  Every lane
// of the wave must be here.
template <int R>
__device__ __forceinline__ bool tile_div_once(const TileState<R> &s)
{
    bool ok = true;
#pragma unroll
    for (int r = 0; r < R; ++r) ok = ok && div_once_ok(s.gr[r]);
    return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
}

// `iters` primal-dual iterations (src/global_faldoi.cpp:684-783) on the tile held in `s`; shared by the per-launch kernel
// below and by the persistent kernel (tvl1_persist_kernel.hpp).  halo_y = rows of halo above and below the owned rows; the
// caller has put row 0 of every wave's ubar into s_ub1 / s_ub2 and zeroed s_err[0..iters).  Per-lane maxima of |du|^2 over
// the owned pixels are left in s_err[it][lane].
struct TvNoHook {
    __device__ __forceinline__ void operator()(int) const {}
};
// `after_duals(it)`: called by every thread of a computing wave once the duals of iteration `it` are final (before the barrier that
// precedes the primal step) -- the persistent kernel publishes the ring's duals from there in the last iteration of a chunk.
// `after_primal(it)`: the same once flow and ubar of iteration `it` are final.
// ONCE: the quotients by |grad I|^2 and theta in the one-product form (faldoi_device.hpp: div_once).  A property of the whole solve,
// chosen by the host; the kernels that compute warp constants report divisors that do not admit it (TvConst / PersistArgs: unsafe) and
// the host then repeats the solve with ONCE = false.  (A wave-uniform choice inside the kernel -- per quotient, per phase, or two copies
// of this loop -- was built and measured: the branch and the second copy's registers cost what the shorter form saves.)
template <int R, int NW, int BORDER, bool ONCE, typename Hook = TvNoHook, typename Hook2 = TvNoHook>
__device__ __forceinline__ void tvl1_iterate(TileState<R> &s, int iters, int halo_y, int gx, int gy0, int nx, int ny,
                                             int lane, int wv, bool wave_live, const TvConst &prm,
                                             float (*s_ub1)[kTileW], float (*s_ub2)[kTileW], float (*s_x12)[kTileW],
                                             float (*s_x22)[kTileW], unsigned (*s_err)[kTileW], Hook after_duals = Hook(),
                                             Hook2 after_primal = Hook2())
{
    constexpr int TH = R * NW;
    // BORDER bit 0: the tile footprint contains image column 0 or nx-1; bit 1: row 0 or ny-1.  A tile on the left or right
    // edge only (most border tiles) keeps the row formulas of the interior, and vice versa.
    constexpr bool COLB = (BORDER & 1) != 0, ROWB = (BORDER & 2) != 0;
    const bool last_c = COLB && (gx == nx - 1);
    const bool first_c = COLB && (gx == 0);

    // (recording max |du|^2 only in the iterations a static schedule reports, with the loop body compiled twice, was measured:
    // 2.4 % SLOWER than recording it always -- the doubled body costs more than the eight instructions per pixel it saves)
    for (int it = 0; it < iters; ++it) {
        // Rows whose values can still reach the owned rows [halo_y, TH-halo_y) before the launch ends:
        // after this iteration u is needed on tile rows [need_lo, need_hi), the new duals one row
        // further up (the divergence reads xi12/xi22 of the row above).  Everything else is skipped
        // (scalar tests: tile row = wv*R + r is wave-uniform).
        const int reach = iters - 1 - it;
        const int need_lo = halo_y - reach, need_hi = TH - halo_y + reach;
        if (!(kWhatIf & 1)) __syncthreads();                   // ubar row 0 of every wave visible
        // ---- dual step: forward gradient of ubar (:725-726) + ofTVl2_getD (:735) ----
        if (wave_live && wv * R + R > need_lo - 1 && wv * R < need_hi) {
            const float dn1 = (wv < NW - 1) ? s_ub1[wv < NW - 1 ? wv + 1 : wv][lane] : 0.0f;
            const float dn2 = (wv < NW - 1) ? s_ub2[wv < NW - 1 ? wv + 1 : wv][lane] : 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (wv * R + r < need_lo - 1 || wv * R + r >= need_hi) continue;
                // neighbours are fetched inside the executed block so the DPP moves fold into the
                // subtractions (v_sub_f32_dpp)
                const float r1 = lane_right(s.ub1[r]);
                const float r2 = lane_right(s.ub2[r]);
                const float d1 = (r < R - 1) ? s.ub1[r < R - 1 ? r + 1 : r] : dn1;
                const float d2 = (r < R - 1) ? s.ub2[r < R - 1 ? r + 1 : r] : dn2;
                const bool last_r = ROWB && (gy0 + r == ny - 1);
                dual_step(s.x11[r], s.x12[r], s.x21[r], s.x22[r],
                          fwd_diff(r1, s.ub1[r], last_c), fwd_diff(d1, s.ub1[r], last_r),
                          fwd_diff(r2, s.ub2[r], last_c), fwd_diff(d2, s.ub2[r], last_r), prm.tau);
            }
            s_x12[wv][lane] = s.x12[R - 1];
            s_x22[wv][lane] = s.x22[R - 1];
            after_duals(it);
        }
        if (!(kWhatIf & 1)) __syncthreads();                   // new duals' last rows visible
        // ---- primal step: divergence (:745-746), TH (:690-718), getP (:770), relax (:780) ----
        if (wave_live && wv * R + R > need_lo && wv * R < need_hi) {
            const float up12 = (wv > 0) ? s_x12[wv > 0 ? wv - 1 : 0][lane] : 0.0f;
            const float up22 = (wv > 0) ? s_x22[wv > 0 ? wv - 1 : 0][lane] : 0.0f;
            float emax = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int gy = gy0 + r;
                if (wv * R + r < need_lo || wv * R + r >= need_hi) continue;
                const float l11 = lane_left(s.x11[r]);
                const float l21 = lane_left(s.x21[r]);
                const float t12 = (r > 0) ? s.x12[r > 0 ? r - 1 : 0] : up12;
                const float t22 = (r > 0) ? s.x22[r > 0 ? r - 1 : 0] : up22;
                float div1, div2;
                if (BORDER) {
                    const PixClass pc{ first_c, last_c, ROWB && gy == 0, ROWB && gy == ny - 1 };
                    div1 = div_sample(s.x11[r], l11, s.x12[r], t12, pc);
                    div2 = div_sample(s.x21[r], l21, s.x22[r], t22, pc);
                } else {                                       // src/utils.cpp:250-256
                    div1 = (s.x11[r] - l11) + (s.x12[r] - t12);
                    div2 = (s.x21[r] - l21) + (s.x22[r] - t22);
                }
                float v1, v2;
                threshold_step<ONCE>(s.rc[r], s.ix[r], s.iy[r], s.gr[r], s.rgr[r], s.u1[r], s.u2[r], prm.l_t, v1, v2);
                const float e = primal_step<ONCE>(s.u1[r], s.u2[r], s.ub1[r], s.ub2[r], v1, v2, div1, div2, prm.theta,
                                                  prm.rtheta, prm.tau);
                emax = (s.own[r] && e > emax) ? e : emax;
            }
            s_ub1[wv][lane] = s.ub1[0];
            s_ub2[wv][lane] = s.ub2[0];
            // per-lane running maximum of this iteration (values >= 0: uint order == float order;
            // max is order-independent => deterministic); reduced across lanes once, at the end
            atomicMax(&s_err[it][lane], __float_as_uint(emax));
            after_primal(it);
        }
    }
}

// Wave i reduces iteration i's 64 per-lane maxima and stores them (plain stores, read by the next kernel of the stream).
template <int NW>
__device__ __forceinline__ void tvl1_store_errors(unsigned (*s_err)[kTileW], int iters, int lane, int wv,
                                                  float *__restrict__ err_slot_base)
{
    for (int i = wv; i < kMaxChunk; i += NW) {
        float m = (i < iters) ? __uint_as_float(s_err[i][lane]) : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float other = __shfl_xor(m, o);
            m = other > m ? other : m;
        }
        if (lane == 0) err_slot_base[i] = m;
    }
}

template <int R, int NW, int BORDER, bool ONCE>
__device__ __forceinline__ void tvl1_chunk_body(
    float *__restrict__ base, size_t plane_elems, int nx, int ny, int npairs, int ntiles, int t,
    int pair, int txi, int tyi, int halo, const TvConst &prm, float *__restrict__ err_out,
    int cur, int first /* bit 0: starts a warp, bit 1: ends one */, int iters, int err_slot,
    float (*s_ub1)[kTileW], float (*s_ub2)[kTileW], float (*s_x12)[kTileW], float (*s_x22)[kTileW],
    unsigned (*s_err)[kTileW], unsigned long long *stamps)
{
    constexpr int TH = R * NW;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index, scalar
    const int Sx = kTileW - 2 * halo, Sy = TH - 2 * halo;
    const int gx = txi * Sx - halo + lane;
    const int gy0 = tyi * Sy - halo + wv * R;   // scalar: every row test below is wave-uniform
    const bool col_in = gx >= 0 && gx < nx;
    const bool col_own = col_in && gx >= txi * Sx && gx < (txi + 1) * Sx;
    // a wave whose rows all lie outside the image only takes part in the barriers
    const bool wave_live = (gy0 + R > 0) && (gy0 < ny);

    float *slab = base + (size_t)pair * P_COUNT * plane_elems;
    const float *src = slab + (size_t)(cur ? P_STATE1 : P_STATE0) * plane_elems;
    float *dst = slab + (size_t)(cur ? P_STATE0 : P_STATE1) * plane_elems;
    const float *rcp = slab + (size_t)P_RC * plane_elems;
    const float *iwxp = slab + (size_t)P_IWX * plane_elems;
    const float *iwyp = slab + (size_t)P_IWY * plane_elems;

    TileState<R> s;
    bool in[R];

#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int gy = gy0 + r;
        in[r] = col_in && gy >= 0 && gy < ny;
        s.own[r] = in[r] && col_own && gy >= tyi * Sy && gy < (tyi + 1) * Sy;
        const int p = in[r] ? gy * nx + gx : 0;
        s.u1[r] = in[r] ? src[0 * plane_elems + p] : 0.0f;
        s.u2[r] = in[r] ? src[1 * plane_elems + p] : 0.0f;
        s.x11[r] = in[r] ? src[4 * plane_elems + p] : 0.0f;
        s.x12[r] = in[r] ? src[5 * plane_elems + p] : 0.0f;
        s.x21[r] = in[r] ? src[6 * plane_elems + p] : 0.0f;
        s.x22[r] = in[r] ? src[7 * plane_elems + p] : 0.0f;
        s.rc[r] = in[r] ? rcp[p] : 0.0f;
        s.ix[r] = in[r] ? iwxp[p] : 0.0f;
        s.iy[r] = in[r] ? iwyp[p] : 0.0f;
        s.gr[r] = grad_sq(s.ix[r], s.iy[r]);                    // src/global_faldoi.cpp:648-652
        s.rgr[r] = ONCE ? recip_f64_full(s.gr[r]) : recip_f64(s.gr[r]);
    }
    const bool ends_warp = (first & 2) != 0;                    // the launch surely ends its warp: ubar is dead afterwards
    first &= 1;
    if (first) {                                                // start of a warp: ubar = u (:659-660)
#pragma unroll
        for (int r = 0; r < R; ++r) { s.ub1[r] = s.u1[r]; s.ub2[r] = s.u2[r]; }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = in[r] ? (gy0 + r) * nx + gx : 0;
            s.ub1[r] = in[r] ? src[2 * plane_elems + p] : 0.0f;
            s.ub2[r] = in[r] ? src[3 * plane_elems + p] : 0.0f;
        }
    }

    // measurement aid (FALDOI_STAMPS=1, never in a timed run): when do the loads land, the iterations end, the stores drain
    if (stamps && threadIdx.x == 0) {
        float sink = s.u1[R - 1] + s.x22[R - 1] + s.rc[R - 1] + s.iy[R - 1];
        asm volatile("" : "+v"(sink));                                // all loads of this wave have landed
        stamps[4 * t + 1] = __builtin_amdgcn_s_memrealtime();
    }
    for (int i = wv; i < kMaxChunk; i += NW) s_err[i][lane] = 0u;
    s_ub1[wv][lane] = s.ub1[0];
    s_ub2[wv][lane] = s.ub2[0];

    // (storing the four dual planes from inside the last iteration -- they are final after its dual step -- was measured here too:
    // no gain, 13.85 against 13.82 ms per solve; plain stores at a kernel's end stall nobody, unlike the persistent kernel's drain)
    tvl1_iterate<R, NW, BORDER, ONCE>(s, iters, halo, gx, gy0, nx, ny, lane, wv, wave_live, prm, s_ub1, s_ub2, s_x12, s_x22, s_err);

    if (stamps && threadIdx.x == 0) stamps[4 * t + 2] = __builtin_amdgcn_s_memrealtime();
    // ---- store the owned interior into the other state buffer ----
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (s.own[r]) {
            const int p = (gy0 + r) * nx + gx;
            dst[0 * plane_elems + p] = s.u1[r];
            dst[1 * plane_elems + p] = s.u2[r];
            if (!ends_warp) {                                   // the next warp starts from ubar = u (:659-660)
                dst[2 * plane_elems + p] = s.ub1[r];
                dst[3 * plane_elems + p] = s.ub2[r];
            }
            dst[4 * plane_elems + p] = s.x11[r];
            dst[5 * plane_elems + p] = s.x12[r];
            dst[6 * plane_elems + p] = s.x21[r];
            dst[7 * plane_elems + p] = s.x22[r];
        }
    }
    __syncthreads();
    if (stamps && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[4 * t + 3] = __builtin_amdgcn_s_memrealtime();
    }
    tvl1_store_errors<NW>(s_err, iters, lane, wv, err_out + ((size_t)err_slot * ntiles * npairs + t) * kMaxChunk);
}

template <int R, int NW, bool ONCE>
__global__ __launch_bounds__(64 * NW) void tvl1_chunk_kernel(
    float *__restrict__ base, size_t plane_elems, int nx, int ny, int ntx, int nty, int npairs,
    int halo, TvConst prm, const PairCtl *__restrict__ ctl, float *__restrict__ err_out,
    int static_mode, int s_cur, int s_first, int s_iters, int err_slot, unsigned long long *stamps)
{
    constexpr int TH = R * NW;
    __shared__ float s_ub1[NW][kTileW];
    __shared__ float s_ub2[NW][kTileW];
    __shared__ float s_x12[NW][kTileW];
    __shared__ float s_x22[NW][kTileW];
    __shared__ unsigned s_err[kMaxChunk][kTileW];

    const int per = gridDim.x >> 3;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int ntiles = ntx * nty;
    if (t >= ntiles * npairs) return;
    if (stamps && threadIdx.x == 0) stamps[4 * t] = __builtin_amdgcn_s_memrealtime();
    const int pair = t / ntiles;
    const int tile = t - pair * ntiles;
    const int tyi = tile / ntx, txi = tile - tyi * ntx;

    int cur, first, iters;
    if (static_mode) {
        cur = s_cur; first = s_first; iters = s_iters;
    } else {
        const PairCtl c = ctl[pair];
        cur = c.cur; iters = c.iters;
        // bit 1: this launch ends the warp whatever the tolerance test says (exact-length re-run, or the iteration bound)
        first = (c.first ? 1 : 0) | ((c.redo || c.n + c.iters >= prm.max_inner) ? 2 : 0);
    }
    if (iters <= 0) return;                     // uniform across the block

    // does the tile footprint [X0, X0+64) x [Y0, Y0+TH) touch the image frame?
    const int X0 = txi * (kTileW - 2 * halo) - halo, Y0 = tyi * (TH - 2 * halo) - halo;
    const bool colb = X0 <= 0 || X0 + kTileW >= nx, rowb = Y0 <= 0 || Y0 + TH >= ny;
#define FALDOI_BODY(B)                                                                                                  \
    tvl1_chunk_body<R, NW, B, ONCE>(base, plane_elems, nx, ny, npairs, ntiles, t, pair, txi, tyi, halo, prm, err_out, \
                                    cur, first, iters, err_slot, s_ub1, s_ub2, s_x12, s_x22, s_err, stamps)
    if (colb && rowb) FALDOI_BODY(3);
    else if (colb) FALDOI_BODY(1);
    else if (rowb) FALDOI_BODY(2);
    else FALDOI_BODY(0);
#undef FALDOI_BODY
}

// ---------------------------------------------------------------------------------------------
// Control kernel (dynamic mode): one block per pair, run after every chunk launch.
// Reduces the per-tile error maxima and advances the pair's state machine exactly like the
// reference's loop control  `while (err_D > tol*tol && n < MAX_ITERATIONS_GLOBAL)`
// (src/global_faldoi.cpp:684) and `for warpings < warps` (:631).
// iters_out/err_out: [npairs][warps] -- what the reference prints at :803.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tvl1_ctl_kernel(PairCtl *__restrict__ ctl, const float *__restrict__ err_part,
                                                       int ntiles, TvConst prm, int *__restrict__ iters_out,
                                                       float *__restrict__ err_out, int *__restrict__ done_count)
{
    __shared__ float s_e[kMaxChunk][256];
    const int pair = blockIdx.x;
    PairCtl c = ctl[pair];
    if (c.done || c.iters <= 0) return;

    // e[i] = max over this pair's tiles of the tile maxima (values are >= 0; max is exact)
    float loc[kMaxChunk];
#pragma unroll
    for (int i = 0; i < kMaxChunk; ++i) loc[i] = 0.0f;
    const float *ep = err_part + (size_t)pair * ntiles * kMaxChunk;
    for (int tl = threadIdx.x; tl < ntiles; tl += blockDim.x) {
#pragma unroll
        for (int i = 0; i < kMaxChunk; ++i) {
            const float v = ep[(size_t)tl * kMaxChunk + i];
            loc[i] = v > loc[i] ? v : loc[i];
        }
    }
#pragma unroll
    for (int i = 0; i < kMaxChunk; ++i) s_e[i][threadIdx.x] = loc[i];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int i = 0; i < kMaxChunk; ++i) {
                const float a = s_e[i][threadIdx.x], b = s_e[i][threadIdx.x + s];
                s_e[i][threadIdx.x] = b > a ? b : a;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;

    // first iteration of this chunk after which the reference's while-condition fails
    int stop_at = -1;
    if (prm.tol_stop && !c.redo) {
        for (int i = 0; i < c.iters; ++i)
            if (!(s_e[i][0] > prm.tol2)) { stop_at = i; break; }
    }
    bool warp_ends = false;
    if (stop_at >= 0 && stop_at < c.iters - 1) {
        // stopped inside the chunk: re-run the same input for exactly stop_at+1 iterations
        c.redo = 1;
        c.iters = stop_at + 1;
    } else {
        c.n += c.iters;
        const float e_last = s_e[c.iters - 1][0];
        warp_ends = c.redo || stop_at >= 0 || c.n >= prm.max_inner;
        c.cur ^= 1;
        c.redo = 0;
        if (warp_ends) {
            iters_out[pair * prm.warps + c.warp] = c.n;
            err_out[pair * prm.warps + c.warp] = e_last;
            c.warp += 1;
            c.n = 0;
            c.first = 1;
        } else {
            c.first = 0;
        }
        if (c.warp >= prm.warps) {
            c.done = 1;
            c.iters = 0;
            atomicAdd(done_count, 1);
        } else {
            const int left = prm.max_inner - c.n;
            c.iters = left < prm.chunk ? left : prm.chunk;
        }
    }
    ctl[pair] = c;
}

// Static mode epilogue: per-warp error = max over tiles of the last iteration's tile maxima.
// err_part: [warps][npairs*ntiles][kMaxChunk]; last_iters = iterations of the warp's last chunk.
__global__ __launch_bounds__(256) void tvl1_static_stats_kernel(const float *__restrict__ err_part, int ntiles,
                                                                int npairs, int warps, int max_inner, int last_iters,
                                                                int *__restrict__ iters_out, float *__restrict__ err_out)
{
    __shared__ float s_e[256];
    const int pair = blockIdx.x % npairs;
    const int w = blockIdx.x / npairs;
    const float *ep = err_part + ((size_t)w * npairs + pair) * ntiles * kMaxChunk;
    float m = 0.0f;
    for (int tl = threadIdx.x; tl < ntiles; tl += blockDim.x) {
        const float v = ep[(size_t)tl * kMaxChunk + (last_iters - 1)];
        m = v > m ? v : m;
    }
    s_e[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const float b = s_e[threadIdx.x + s];
            if (b > s_e[threadIdx.x]) s_e[threadIdx.x] = b;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        iters_out[pair * warps + w] = max_inner;
        err_out[pair * warps + w] = s_e[0];
    }
}

}  // namespace faldoi
