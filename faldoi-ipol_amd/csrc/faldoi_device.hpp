// faldoi_device.hpp -- per-pixel device functions of the FALDOI TV-L1 path (gfx950).
//
// Everything here is a pure function of values held in registers; the kernels in
// tvl1_kernels.hpp decide where those values come from (HBM, LDS, a neighbouring lane).
// Parity contract: each function evaluates the reference's floating-point expression tree
// (file:line cited, upstream repo fperezgamonal/faldoi-ipol) with the same association and
// the same precision per operation.  The translation unit is compiled with
// -ffp-contract=off (no fused multiply-add), hipcc's default correctly-rounded fp32
// divide/sqrt and IEEE denormals, so results are bit-identical to the CPU path.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

// FALDOI_WHATIF (tools/whatif_probe.py, never defined in the product build): timing-only builds with one ingredient removed
// -- wrong results, same instruction stream otherwise -- to see what a kernel's time is made of.
//   1 no workgroup barriers inside an iteration   2 the duals' quotient path never taken   4 threshold quotient -> product
//   8 primal quotients by theta -> products       16 no waiting for the neighbours           32 no warp constants
#ifndef FALDOI_WHATIF
#define FALDOI_WHATIF 0
#endif

namespace faldoi {

constexpr int kWhatIf = FALDOI_WHATIF;

// Smallest float that is NOT below the reference's double threshold GRAD_IS_ZERO = 1E-8
// (src/parameters.h:45): for float g, ((double)g < 1E-8) == (g < kGradZeroF).
// (float)1e-8 = 0x322bcc77 = 9.99999994e-09 < 1e-8, so the bound is the next float up.
__device__ __forceinline__ float grad_zero_bound() { return __uint_as_float(0x322bcc78u); }

// ---------------------------------------------------------------------------------------------
// Correctly rounded fp32 quotients through fp64 (gfx950: an IEEE fp32 divide costs ~40 SIMD
// cycles, cvt+mul+2 fma+cvt in fp64 ~21, and one reciprocal serves several numerators).
//
//   rd  ~ 1/d in fp64, relative error <= 2^-27 suffices (recip_f64: v_rcp_f64 + one Newton step)
//   q0  = a * rd                       (fp64)
//   rem = a - d*q0  by one fma         (fp64, one rounding of an almost-cancelled sum)
//   q1  = q0 + rem*rd  by one fma      error <= 2^-52 relative (Markstein correction)
//   result = (float) q1                RN to fp32
//
// Why the last rounding gives RN(a/d): for fp32 a, d the exact quotient is never closer than
// 2^-49 (relative) to a midpoint of two fp32 values unless it IS exactly representable in fp64
// (ties only exist for subnormal fp32 results); in the first case an fp64 value within 2^-52
// rounds the same way, in the second the correction step returns that fp64 value exactly and
// the hardware conversion breaks the tie to even like the IEEE divide.  Signed zeros, subnormal
// operands and subnormal results follow IEEE semantics because every step is an IEEE fp64
// operation; NaN operands give NaN.  Domain: finite a, finite d > 0 (call sites: d = max(1,|xi|)
// and d = theta > 0); an infinite numerator yields NaN instead of infinity -- only reachable
// after the iteration has already overflowed.  Checked against the hardware IEEE divide on the
// GPU over random bit patterns and edge cases by faldoi_selftest_division (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double recip_f64(float d)
{
    const double dd = (double)d;
    const double r0 = __builtin_amdgcn_rcp(dd);                // v_rcp_f64: full fp32 range, >= 2^-20
    const double e = __builtin_fma(-dd, r0, 1.0);
    return __builtin_fma(e, r0, r0);                           // >= 2^-40
}
__device__ __forceinline__ float div_by(float a, float d, double rd)
{
    const double ad = (double)a, dd = (double)d;
    const double q0 = ad * rd;
    // negated residual times negated reciprocal: the same value as fma(a - d*q0, rd, q0), but an
    // exactly cancelled residual (+0) then contributes -0, so a = -0 keeps its sign (d, rd > 0)
    const double nrem = __builtin_fma(dd, q0, -ad);
    return (float)__builtin_fma(nrem, -rd, q0);
}

// ---------------------------------------------------------------------------------------------
// The same quotient as ONE fp64 product, for divisors that are constant over many quotients (theta: a solve; |grad I|^2: a warp):
//
//   rd  = 1/d to within 2^-52 relative (recip_f64_full: two Newton steps; on the host 1.0 / (double)d)
//   result = (float)((double)a * rd)                 cvt + mul + cvt instead of cvt + mul + 2 fma + cvt
//
// q0 = a * rd is within 2^-52 + 2^-53 of a/d (relative).  (float)q0 == RN(a/d) unless a boundary of the fp32 rounding (the midpoint of
// two neighbouring floats, 2^-150, or the overflow threshold) lies between q0 and a/d.  With d = D * 2^f, a = A * 2^g (D, A < 2^24):
//  * result in the normal range or overflowing: a boundary m has 25 significant bits, so a - d*m is a non-zero multiple of
//    2^min(g, f + e - 24) (D*m keeps an odd part of >= 25 bits: it is never a float) and |a/d - m| >= 2^-49 |a/d| -- eight times the
//    error of q0.  A quotient that is exactly a float is rounded to itself.
//  * result subnormal (or rounding to zero): the boundaries are the odd multiples of 2^-150.  a - d*m is a multiple of
//    2^min(g, f - 150); if it is not zero, |a/d - m| >= 2^-48 |a/d| (f - 150 <= g: at least 2^-150 / D against |a/d| < 2^-126;
//    otherwise at least 2^g = |a| / A).  It IS zero -- an exact tie, which RN breaks to even and a perturbed q0 may not -- only if
//    d * (2k+1) * 2^-150 is a float of that magnitude: d = (odd) * 2^f with f >= 1, i.e. d is an even integer; and when its odd
//    part is 1 (a power of two) rd and q0 are exact and the conversion breaks the tie itself.
// Hence div_once_ok(d): everything except the even integers >= 6.  Callers test it once per divisor (a wave-uniform flag) and use div_by
// for the others.  Signed zeros, NaN and subnormal numerators behave as in div_by; an infinite numerator gives the IEEE infinity.
// Checked against the hardware IEEE divide by faldoi_selftest_division: random bit patterns and tie makers for random divisors, and
// EVERY numerator for theta = 0.3f.  (Measured: 11.85 -> 11.5 ms per 300 x 10 solve.  For the duals -- a new divisor per pixel and
// iteration -- the two extra Newton steps and the per-pixel test cost more than the two fma per quotient they save: not used there.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double recip_f64_full(float d)
{
    const double dd = (double)d;
    const double r0 = __builtin_amdgcn_rcp(dd);
    const double e0 = __builtin_fma(-dd, r0, 1.0);
    const double r1 = __builtin_fma(e0, r0, r0);               // >= 2^-40
    const double e1 = __builtin_fma(-dd, r1, 1.0);
    return __builtin_fma(e1, r1, r1);                          // (1/d)(1 - e1^2) rounded once: <= 2^-53 + 2^-79
}
__device__ __forceinline__ bool div_once_ok(float d)           // false: d is an even integer >= 6 (every float >= 2^24 is one)
{
    return !(d >= 6.0f) || __builtin_amdgcn_fractf(0.5f * d) != 0.0f;
}
__device__ __forceinline__ float div_once(float a, double rd)
{
    return (float)((double)a * rd);
}
// a / d by whichever form the caller has established for its divisors.  ONCE is a compile-time choice: the callers branch once per
// phase on a wave-uniform flag and instantiate both bodies (a branch per quotient was measured: it costs what the shorter form saves).
template <bool ONCE>
__device__ __forceinline__ float div_const(float a, float d, double rd)
{
    return ONCE ? div_once(a, rd) : div_by(a, d, rd);
}

// (Measured and not adopted: x / theta as fma(x, zh, x*zl) with zh + zl ~ 1/theta -- Brisebarre, Muller, Raina 2004 -- proved per
// theta by an exhaustive comparison over all 2^32 numerators; theta = 0.3 passes for |x| >= 2^-100.  Unguarded it saves 2.3 % of a
// solve; the wave-uniform guard for tiny non-zero numerators costs the same 2.3 %.)

// Neighbouring lanes of the wavefront (tile column x-1 / x+1) through the DPP crossbar.
// Lanes without a source (63 / 0) read 0 (bound_ctrl); the callers never use it (halo).  old = 0
// with bound_ctrl lets the compiler fold the move into the consuming VOP2 (v_sub_f32_dpp).
__device__ __forceinline__ float lane_right(float v)     // value held by lane+1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v),
                                                                0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_left(float v)      // value held by lane-1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v),
                                                                0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}

// Correctly rounded square root for x in [1, +inf] (and NaN): v_sqrt_f32 is within 1 ulp, so the
// result is one of {y-1ulp, y, y+1ulp}; the two fused residuals x - c*y pick it.  This is the
// correction hipcc itself emits for sqrtf, without the input scaling / zero / subnormal handling
// that a general argument needs (here x >= 1).  Checked exhaustively on the GPU against sqrtf for
// every float in [1, FLT_MAX] by faldoi_selftest_division.
__device__ __forceinline__ float sqrt_ge1(float x)
{
    const float y = __builtin_amdgcn_sqrtf(x);
    const float yd = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    const float yu = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    const float rd = __builtin_fmaf(-yd, y, x);
    const float ru = __builtin_fmaf(-yu, y, x);
    float r = (rd <= 0.0f) ? yd : y;
    r = (ru > 0.0f) ? yu : r;
    return r;
}

// The reference's |g| written as sqrt(g*g) (src/global_faldoi.cpp:1151).  In radix 2 with round-to-
// nearest, sqrt(RN(g*g)) == |g| exactly whenever g*g neither underflows nor overflows (and for
// g == 0); outside that range the IEEE sequence is evaluated.  The branch is wave-uniform.  The
// identity is checked on the GPU for every float of the fast range by faldoi_selftest_division.
__device__ __forceinline__ bool sqrt_of_square_is_abs(float g)
{
    const float a = __builtin_fabsf(g);
    return (a >= 0x1p-60f && a <= 0x1p60f) || a == 0.0f;
}
__device__ __forceinline__ float sqrt_of_square(float g)
{
    if (__builtin_amdgcn_ballot_w64(!sqrt_of_square_is_abs(g)) == 0ull) return __builtin_fabsf(g);
    // The empty asm keeps this a BRANCH: without it the compiler if-converts the wave-uniform test and evaluates the ~20
    // instructions of the IEEE root for every value, selecting afterwards (measured: 22 % of the NLTV dual step's instructions).
    asm volatile("" ::: "memory");
    return __builtin_sqrtf(g * g);
}

// ---------------------------------------------------------------------------------------------
// Stencil classes of a pixel (its position relative to the image frame).
// ---------------------------------------------------------------------------------------------
struct PixClass {
    bool first_c, last_c, first_r, last_r;
};
__device__ __forceinline__ PixClass classify(int x, int y, int nx, int ny)
{
    return PixClass{ x == 0, x == nx - 1, y == 0, y == ny - 1 };
}

// src/utils.cpp:367-423 centered_gradient: 0.5*(next - prev); on the frame the missing side is
// replaced by the pixel itself (one-sided, still halved).  The subtraction is fp32, the product
// with the double literal 0.5 is fp64, stored as fp32 (exact: halving never rounds except for
// denormals, where the fp64 product is still exact and rounds once).
__device__ __forceinline__ float centered_diff(float next, float prev)
{
    const float d = next - prev;
    return (float)(0.5 * (double)d);
}

// src/utils.cpp:285-323 forward_gradient: zero on the last column / last row.
__device__ __forceinline__ float fwd_diff(float next, float self, bool is_last)
{
    return is_last ? 0.0f : next - self;
}

// src/utils.cpp:239-283 divergence, all nine border classes.
// a = v1[p], al = v1[p-1], b = v2[p], bu = v2[p-nx]; values whose class does not use them
// may be anything (they are never touched by the selected expression's result).
__device__ __forceinline__ float div_sample(float a, float al, float b, float bu, const PixClass c)
{
    const bool inner_c = !c.first_c && !c.last_c;
    // horizontal part: interior v1[p]-v1[p-1]; first column v1[p]; last column -v1[p-1]
    const float h = inner_c ? (a - al) : (c.first_c ? a : -al);
    // rows: first row adds v2[p] (:263,:278,:279); last row subtracts v2[p-nx] (:264,:280,:281)
    const float r_first = h + b;
    const float r_last = h - bu;
    // middle rows: interior columns v1x + (v2[p]-v2[p-nx]) (:250-256); frame columns are written
    // left-to-right in the reference, (h + v2[p]) - v2[p-nx] (:272-273)
    const float r_mid = inner_c ? h + (b - bu) : (h + b) - bu;
    return c.first_r ? r_first : (c.last_r ? r_last : r_mid);
}

// ---------------------------------------------------------------------------------------------
// Bicubic interpolation, src/bicubic_interpolation.c:26-40,103-237.
// ---------------------------------------------------------------------------------------------
struct BicubicTaps {
    int X[4];      // tap columns in the reference's order: minus, centre, plus, plus-plus
    int Y[4];      // tap rows, same order
    float tx, ty;  // fractional offsets (uu - x, vv - y) against the CLAMPED centre
    bool clamped;  // any index was clamped ("out" in the reference)
};

__device__ __forceinline__ int clamp_flag(int i, int n, bool &clamped)
{
    if (i < 0) { clamped = true; return 0; }
    if (i >= n) { clamped = true; return n - 1; }
    return i;
}

// :138-166.  (int) truncates toward zero; the "minus" row uses sx, not sy (reference quirk, :159).
__device__ __forceinline__ BicubicTaps bicubic_taps(float uu, float vv, int nx, int ny)
{
    BicubicTaps t;
    const int sx = (uu < 0) ? -1 : 1;
    const int sy = (vv < 0) ? -1 : 1;
    const int iu = (int)uu, iv = (int)vv;
    t.clamped = false;
    const int x = clamp_flag(iu, nx, t.clamped);
    const int y = clamp_flag(iv, ny, t.clamped);
    t.X[0] = clamp_flag(iu - sx, nx, t.clamped);
    t.Y[0] = clamp_flag(iv - sx, ny, t.clamped);
    t.X[1] = x;
    t.Y[1] = y;
    t.X[2] = clamp_flag(iu + sx, nx, t.clamped);
    t.Y[2] = clamp_flag(iv + sy, ny, t.clamped);
    t.X[3] = clamp_flag(iu + 2 * sx, nx, t.clamped);
    t.Y[3] = clamp_flag(iv + 2 * sy, ny, t.clamped);
    t.tx = uu - (float)x;
    t.ty = vv - (float)y;
    return t;
}

// :103-111 cubic_interpolation_cell.  (v2-v0) and (v1-v2) are fp32 subtractions; everything
// that touches a double literal is fp64; one rounding to fp32 at the end.
// The reference evaluates, without contraction,
//     c3 = 3.0*d12 + v3 - v0;   c2 = 2.0*v0 - 5.0*v1 + 4.0*v2 - v3 + t*c3;   c1 = d20 + t*c2;   v1 + 0.5*t*c1.
// A product of a float by 2, 3, 4 or 5 is exact in double (<= 27 significant bits), and fma(a, b, c) with an
// exact a*b rounds exactly like (a*b) + c: those products are fused (13 fp64 operations instead of 16);
// t*c3, t*c2 and (0.5*t)*c1 are inexact products and stay separate.
__device__ __forceinline__ float keys_cubic(float v0, float v1, float v2, float v3, float t)
{
    const float d20 = v2 - v0;
    const float d12 = v1 - v2;
    const double td = (double)t;
    const double e0 = (double)v0, e1 = (double)v1, e2 = (double)v2, e3 = (double)v3;
    const double c3 = __builtin_fma(3.0, (double)d12, e3) - e0;
    const double c2 = (__builtin_fma(4.0, e2, __builtin_fma(-5.0, e1, 2.0 * e0)) - e3) + td * c3;
    const double c1 = (double)d20 + td * c2;
    return (float)(e1 + 0.5 * td * c1);
}

// :113-131,:199-236: 16 gathers, four column cubics along y, one cubic along x.
__device__ __forceinline__ float bicubic_eval(const float *__restrict__ img, const BicubicTaps &t, int nx)
{
    float col[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float p0 = img[t.X[a] + nx * t.Y[0]];
        const float p1 = img[t.X[a] + nx * t.Y[1]];
        const float p2 = img[t.X[a] + nx * t.Y[2]];
        const float p3 = img[t.X[a] + nx * t.Y[3]];
        col[a] = keys_cubic(p0, p1, p2, p3, t.ty);
    }
    return keys_cubic(col[0], col[1], col[2], col[3], t.tx);
}

// The same evaluation for a pixel NONE of whose tap indices was clamped: the four tap columns are then the consecutive
// columns x0 .. x0+3 (ascending for sx = +1, descending for sx = -1), so every tap row is ONE 16-byte load instead of
// four 4-byte gathers -- the per-warp kernel is bound by the number of vector-memory instructions (64 lane addresses
// each), not by bytes.  The loads are only 4-byte aligned; gfx950 takes unaligned dwordx4 global loads.
struct __attribute__((packed, aligned(4))) TapRow { float c0, c1, c2, c3; };
__device__ __forceinline__ float bicubic_eval_rows(const float *__restrict__ img, const BicubicTaps &t, int nx)
{
    const bool asc = t.X[0] < t.X[1];
    const int x0 = asc ? t.X[0] : t.X[3];
    const TapRow r0 = *reinterpret_cast<const TapRow *>(img + x0 + nx * t.Y[0]);
    const TapRow r1 = *reinterpret_cast<const TapRow *>(img + x0 + nx * t.Y[1]);
    const TapRow r2 = *reinterpret_cast<const TapRow *>(img + x0 + nx * t.Y[2]);
    const TapRow r3 = *reinterpret_cast<const TapRow *>(img + x0 + nx * t.Y[3]);
    const float k0 = keys_cubic(r0.c0, r1.c0, r2.c0, r3.c0, t.ty);          // column x0 ... x0+3, rows in the reference's order
    const float k1 = keys_cubic(r0.c1, r1.c1, r2.c1, r3.c1, t.ty);
    const float k2 = keys_cubic(r0.c2, r1.c2, r2.c2, r3.c2, t.ty);
    const float k3 = keys_cubic(r0.c3, r1.c3, r2.c3, r3.c3, t.ty);
    return keys_cubic(asc ? k0 : k3, asc ? k1 : k2, asc ? k2 : k1, asc ? k3 : k0, t.tx);
}

// border_out = false call sites (clamped taps are legal there): the row form when no lane of the wavefront has a clamped
// tap, the gather form for the whole wavefront otherwise (near the image frame).  any_clamped is wave-uniform.
__device__ __forceinline__ bool wave_any(bool v) { return __builtin_amdgcn_ballot_w64(v) != 0ull; }
__device__ __forceinline__ float bicubic_eval_any(const float *__restrict__ img, const BicubicTaps &t, int nx, bool any_clamped)
{
    return any_clamped ? bicubic_eval(img, t, nx) : bicubic_eval_rows(img, t, nx);
}

// ---------------------------------------------------------------------------------------------
// TV-L1 per-pixel steps, src/global_faldoi.cpp.
// ---------------------------------------------------------------------------------------------

// :647-657 per-warp constants.
__device__ __forceinline__ float rho_const(float I1w, float I1wx, float I1wy, float u1, float u2, float I0)
{
    return I1w - I1wx * u1 - I1wy * u2 - I0;
}
__device__ __forceinline__ float grad_sq(float I1wx, float I1wy)
{
    const float Ix2 = I1wx * I1wx;
    const float Iy2 = I1wy * I1wy;
    return Ix2 + Iy2;
}

// :690-718 thresholding step: v = u + TH(rho, grad).
// rgrad = recip_f64(grad), constant over a launch; the reference's `fi = -rho / grad` is obtained
// as the exact quotient div_by(-rho, grad, rgrad) (grad >= 1e-8 > 0 wherever fi is selected; for
// smaller grad the value is computed but discarded).  ONCE: the caller guarantees rgrad = recip_f64_full(grad)
// and div_once_ok(grad) for every lane -- the one-product form.  Branch-free: one selected factor f with
// d = f * I1w{x,y}; the three explicit factors are the reference's own operands (l_t, -l_t, fi).
template <bool ONCE = false>
__device__ __forceinline__ void threshold_step(float rho_c, float I1wx, float I1wy, float grad, double rgrad,
                                               float u1, float u2, float l_t, float &v1, float &v2)
{
    const float rho = rho_c + (I1wx * u1 + I1wy * u2);
    const float lg = l_t * grad;
    const bool below = rho < -lg;                 // (-l_t)*grad == -(l_t*grad) exactly
    const bool above = rho > lg;
    const float fi = (kWhatIf & 4) ? -rho * (float)rgrad : div_const<ONCE>(-rho, grad, rgrad);
    const float f = below ? l_t : (above ? -l_t : fi);
    float d1 = f * I1wx;
    float d2 = f * I1wy;
    const bool flat = !below && !above && (grad < grad_zero_bound());
    d1 = flat ? 0.0f : d1;                        // `d1 = d2 = 0` (:706), not a product: keeps +0
    d2 = flat ? 0.0f : d2;
    v1 = u1 + d1;
    v2 = u2 + d2;
}

// :349-381 ofTVl2_getD: xi <- (xi + tau * grad(ubar)) / max(1, |xi_old|).  Step tau (not
// tau/theta) and the norm of the PRE-update duals: reference behaviour, reproduced.
__device__ __forceinline__ void dual_step(float &xi11, float &xi12, float &xi21, float &xi22,
                                          float u1x, float u1y, float u2x, float u2y, float tau)
{
    const float g11 = xi11 * xi11;
    const float g12 = xi12 * xi12;
    const float g21 = xi21 * xi21;
    const float g22 = xi22 * xi22;
    const float g = g11 + g12 + g21 + g22;
    // numerators first, in place (the norm above already used the pre-update duals)
    xi11 = xi11 + tau * u1x;
    xi12 = xi12 + tau * u1y;
    xi21 = xi21 + tau * u2x;
    xi22 = xi22 + tau * u2y;
    // g <= 1  =>  sqrt(g) <= 1 (correctly rounded)  =>  max(1, sqrt(g)) == 1  =>  a / 1 == a.
    // When no lane of the wavefront has g > 1 the square root and the four quotients are skipped
    // as a whole (wave-uniform branch); otherwise every lane takes the general path, which
    // returns the same bits for the g <= 1 lanes.  NaN g (a NaN dual: windows over pruned, unfilled flow): the
    // reference's MAX(1, xi_N) is (1 > xi_N) ? 1 : xi_N, i.e. NaN, and all four duals of the pixel become NaN, not
    // only the NaN ones -- so the tests are written as "not g <= 1" and a NaN g takes the general path as NaN.
    if (!(kWhatIf & 2) && __builtin_amdgcn_ballot_w64(!(g <= 1.0f)) != 0ull) {
        // max(1, sqrt(g)) == sqrt(max(1, g)): sqrt is monotone and sqrt(1) == 1 exactly
        const float n = sqrt_ge1((g <= 1.0f) ? 1.0f : g);
        const double rn = recip_f64(n);             // one reciprocal, four exact quotients
        xi11 = div_by(xi11, n, rn);
        xi12 = div_by(xi12, n, rn);
        xi21 = div_by(xi21, n, rn);
        xi22 = div_by(xi22, n, rn);
    }
}

// :307-342 ofTVl2_getP for one pixel, followed by the over-relaxation of :780-783.
// Returns |du|^2 (u_N).
// rtheta = fp64 reciprocal of theta (host: 1.0 / (double)theta), see div_by; ONCE: div_once_ok(theta) holds, the one-product form.
template <bool ONCE = false>
__device__ __forceinline__ float primal_step(float &u1, float &u2, float &ub1, float &ub2,
                                             float v1, float v2, float div1, float div2,
                                             float theta, double rtheta, float tau)
{
    const float u1k = u1, u2k = u2;
    const float q1 = (kWhatIf & 8) ? (u1k - v1) * (float)rtheta : div_const<ONCE>(u1k - v1, theta, rtheta);
    const float q2 = (kWhatIf & 8) ? (u2k - v2) * (float)rtheta : div_const<ONCE>(u2k - v2, theta, rtheta);
    const float n1 = u1k - tau * (-div1 + q1);
    const float n2 = u2k - tau * (-div2 + q2);
    const float e = (n1 - u1k) * (n1 - u1k) + (n2 - u2k) * (n2 - u2k);
    u1 = n1;
    u2 = n2;
    ub1 = 2.0f * n1 - u1k;
    ub2 = 2.0f * n2 - u2k;
    return e;
}

}  // namespace faldoi
