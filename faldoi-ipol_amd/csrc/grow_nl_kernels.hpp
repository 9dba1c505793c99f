// grow_nl_kernels.hpp -- the other functionals of the local step on the device: of_estimation's branches for
// -m 2..7 (src/energy_model.cpp:691-802) as window solvers of the growing kernel (grow_kernels.hpp).
//
//   M_NLTVL1 (2) / _W (3)    guided_nltvl1[_w]    src/nltv_model.cpp:275, src/nltvw_model.cpp       NL regulariser, L1 data term
//   M_TVCSAD (4) / _W (5)    guided_tvcsad[_w]    src/tvcsad_model.cpp:265, src/tvcsadw_model.cpp   CSAD data term over hypot(|grad|^2, 0.01)
//   M_NLTVCSAD (6)           guided_nltvcsad      src/nltvcsad_model.cpp:297                        NL regulariser + that CSAD term
//   M_NLTVCSAD_W (7)         guided_nltvcsad_w    src/nltvcsadw_model.cpp                           NL + CSAD over sqrt(|grad|^2), GRAD_IS_ZERO guard
//
// One thread per window pixel (<= 121 of 128).  The window is a closed world: a non-local or CSAD neighbour takes
// part only if it lies inside the window (validate_ap_patch, src/aux_energy_model.cpp:33), so everything a call
// needs from other pixels travels through LDS.  State that outlives a call: the 24 + 24 non-local dual values of
// every pixel (HBM planes GrowDir::duals; a call loads the window's, iterates in LDS, stores them back) -- the
// reference never re-initialises them after prepare_stuff.  TV-CSAD keeps nothing: its duals are zeroed per call and
// the flow gradients its dual step reads are never computed (zeros of a fresh allocation), so its
// regulariser is inert and `-div` is -0.
//
// Arithmetic: the model files include <cmath> only (no using-directive), so their unqualified sqrt / fabs / hypot on
// floats are the C double functions: sqrt(x*x) of floats is the correctly rounded float root either way, but the
// products fabs(du) * wp of the energy and every expression with sqrt(grad) in -m 7 run in double.  Divisions are the
// IEEE ones (hipcc's default expansions; this translation unit never uses fast-math).
#pragma once
#include "nltv_kernels.hpp"

#pragma clang fp contract(off)

namespace faldoi {

// (float)hypot((double)g, 0.01), src/tvcsad_model.cpp:337 / src/nltvcsad_model.cpp:357: the C library's double hypot of a
// float-valued g >= 0 and the double 0.01, rounded to float.  Evaluated as the correctly rounded double root of the
// exact sum g^2 + 0.01^2 (g^2 is exact in double; 0.01^2 is carried as a double-double), then one rounding to float.
// glibc's hypot is correctly rounded for all but a vanishing fraction of arguments, and a last-place difference in
// double only matters when it straddles a float rounding boundary (2^-29 of those), see DESIGN.md.
__device__ __forceinline__ float hypot_g_001(float g)
{
    const double y = 0.01;
    const double yy_hi = y * y, yy_lo = __builtin_fma(y, y, -yy_hi);
    const double xx = (double)g * (double)g;                       // exact: 48 significant bits
    const double s_hi = xx + yy_hi;                                // two-sum of xx and yy_hi
    const double bb = s_hi - xx;
    const double s_lo = ((xx - (s_hi - bb)) + (yy_hi - bb)) + yy_lo;
    const double r = __builtin_sqrt(s_hi);
    const double res = __builtin_fma(-r, r, s_hi) + s_lo;          // s - r^2, to ~2^-105
    return (float)(r + res / (2.0 * r));
}

struct NlShared {
    float pq[2 * kNlSlots][kPatchThreads];        // the window's dual values: p slots, then q slots
};

constexpr int kDataL1 = 0, kDataCsad = 1, kDataCsadSqrt = 2;

// Offsets of slot s of a (2R+1)^2 - 1 neighbourhood (rows first, centre skipped).
template <int R> __device__ __forceinline__ void slot_offset(int s, int &dk, int &dl)
{
    const int side = 2 * R + 1, centre = (side * side) / 2;
    const int c = s < centre ? s : s + 1;
    dk = c / side - R;                             // row offset
    dl = c % side - R;                             // column offset
}

// Thread 0 sums sm.t[0 .. m) in index order in fp32 (the reference's serial accumulations).
__device__ __forceinline__ float ordered_sum(const GrowShared &sm, int m)
{
    float acc = 0.0f;
    for (int q = 0; q < m; ++q) acc += sm.t[q];
    return acc;
}

template <bool NL, int DATA>
__device__ __forceinline__ float window_solve_nl(GrowShared &sm, NlShared &ns, const float *__restrict__ base, float *duals,
                                                 const float *__restrict__ wplanes, float *ubase, size_t pe, int nx, int ny,
                                                 int ii, int ij, int ei, int ej, int max_iter_patch, float &u1, float &u2,
                                                 const float *weight, int iiw, int ijw, int wr_call, int method)
{
    const int pw = ei - ii, ph = ej - ij, np = pw * ph, t = threadIdx.x;
    const bool act = t < np;
    const int k = ii + (act ? t % pw : 0), l = ij + (act ? t / pw : 0);
    const int p = l * nx + k;
    const float theta = 0.3f, tau = 0.1f, tol_OF = 0.01f;           // src/energy_model.cpp:714-795
    float lambda = (method == 2) ? 2.0f : 0.85f;
    float gwy = 1.0f, gwx = 1.0f;
    if (weight) {
        const float central = weight[wr_call + 1];
        lambda = lambda / (central * central);
        gwy = weight[l - ij + ijw];
        gwx = weight[k - ii + iiw];
    }
    const float l_t = lambda * theta;
    const float l_t_w = weight ? l_t * gwy * gwx : l_t;
    const float *I0 = base + L_I0 * pe, *I1 = base + L_I1 * pe, *I1x = base + L_I1X * pe, *I1y = base + L_I1Y * pe;

    float i0 = 0, i1w = 0, wx = 0, wy = 0, gr = 0, rc = 0, T1 = 0, T2 = 0, v1 = 0, v2 = 0, ub1 = 0, ub2 = 0, wt = 0;
    double rgrad = 0.0, sq = 1.0;
    bool live = true;
    float w_[NL ? kNlSlots : 1];
    unsigned nlmask = 0;                                            // bit s: non-local slot s has its neighbour inside the window
    unsigned long long csmask = 0ull;                               // bit s: CSAD slot s ...
    if (act) {
        i0 = I0[p];
        const BicubicTaps tp = bicubic_taps((float)k + u1, (float)l + u2, nx, ny);      // border_out = false
        const bool anyc = wave_any(tp.clamped);
        i1w = bicubic_eval_any(I1, tp, nx, anyc);
        wx = bicubic_eval_any(I1x, tp, nx, anyc);
        wy = bicubic_eval_any(I1y, tp, nx, anyc);
        const float g2 = grad_sq(wx, wy);
        if (DATA == kDataL1) {
            gr = g2;
            rc = rho_const(i1w, wx, wy, u1, u2, i0);
            rgrad = recip_f64(gr);
        } else if (DATA == kDataCsad) {
            gr = hypot_g_001(g2);
        } else {
            gr = g2;
            live = (double)gr > 1E-8;                               // GRAD_IS_ZERO, src/nltvcsadw_model.cpp:383,467
            sq = __builtin_sqrt((double)gr);
        }
        T1 = wx * u1;
        T2 = wy * u2;
        ub1 = u1;
        ub2 = u2;
        if (NL) {
#pragma unroll
            for (int s = 0; s < kNlSlots; ++s) {
                int dk, dl;
                slot_offset<kNlR>(s, dk, dl);
                const bool in = k + dl >= ii && k + dl < ei && l + dk >= ij && l + dk < ej;
                nlmask |= in ? (1u << s) : 0u;
                w_[s] = wplanes[(size_t)s * pe + p];
                ns.pq[s][t] = duals[(size_t)s * pe + p];
                ns.pq[kNlSlots + s][t] = duals[(size_t)(kNlSlots + s) * pe + p];
            }
#pragma unroll
            for (int s = 0; s < kNlSlots; ++s)                      // wt: the weights of the in-window neighbours, slot order
                if (nlmask & (1u << s)) wt += w_[s];
        }
        if (DATA != kDataL1) {
#pragma unroll
            for (int s = 0; s < kCsSlots; ++s) {
                int dk, dl;
                slot_offset<kCsR>(s, dk, dl);
                const bool in = k + dl >= ii && k + dl < ei && l + dk >= ij && l + dk < ej;
                csmask |= in ? (1ull << s) : 0ull;
            }
        }
    }
    const int ncs = __builtin_popcountll(csmask);                   // pnei[i].n
    sm.a[t] = ub1;
    sm.b[t] = ub2;
    sm.x[0][t] = i1w;                                               // the warped frame of the whole window (CSAD terms)
    __syncthreads();
    int it = 0;
    float err_D = __builtin_inff();
    while (err_D > tol_OF * tol_OF && it < max_iter_patch) {
        ++it;
        if (act) {
            // ---- data step ----
            if (DATA == kDataL1) {
                threshold_step(rc, wx, wy, gr, rgrad, u1, u2, l_t_w, v1, v2);
            } else {
                v1 = u1;
                v2 = u2;
                if (live) {
                    const float dot = wx * u1 + wy * u2;
                    const float r = (DATA == kDataCsad) ? dot / gr : 0.0f;
                    const double rd = (DATA == kDataCsadSqrt) ? (double)dot / sq : 0.0;
                    float A[kCsSlots];
#pragma unroll
                    for (int s = 0; s < kCsSlots; ++s) {
                        int dk, dl;
                        slot_offset<kCsR>(s, dk, dl);
                        const bool in = (csmask >> s) & 1ull;
                        const int pos = in ? p + dk * nx + dl : p;
                        const int qq = in ? t + dk * pw + dl : t;
                        const float num = i0 - I0[pos] - i1w + sm.x[0][qq] + T1 + T2;
                        float a;
                        if (DATA == kDataCsad) {
                            const float b = num / gr;
                            a = -(b - r);
                        } else {
                            const float b = (float)((double)num / sq);
                            a = (float)(-((double)b - rd));
                        }
                        A[s] = in ? a : __builtin_inff();
                    }
                    sort48(A);
                    float m = __builtin_inff();
#pragma unroll
                    for (int c = 1; c <= kCsSlots; ++c) {
                        const float f = (float)(ncs + 2 - 2 * c) * l_t_w;
                        const float bv = (DATA == kDataCsad) ? f * gr : (float)((double)f * sq);
                        m = __builtin_fminf(m, __builtin_fmaxf(A[c - 1], (c <= ncs + 1) ? bv : __builtin_inff()));
                    }
                    if (DATA == kDataCsad) {
                        v1 = u1 - wx * m / gr;
                        v2 = u2 - wy * m / gr;
                    } else {
                        v1 = (float)((double)u1 - (double)(wx * m) / sq);
                        v2 = (float)((double)u2 - (double)(wy * m) / sq);
                    }
                }
            }
            // ---- dual step on ubar (nltv*_getD): in-window neighbours only ----
            if (NL) {
#pragma unroll
                for (int s = 0; s < kNlSlots; ++s) {
                    int dk, dl;
                    slot_offset<kNlR>(s, dk, dl);
                    const bool in = (nlmask >> s) & 1u;
                    const int qq = in ? t + dk * pw + dl : t;
                    const float nlgr1 = w_[s] * (ub1 - sm.a[qq]) / wt;
                    const float nlgr2 = w_[s] * (ub2 - sm.b[qq]) / wt;
                    const float nl1g = 1 + tau * __builtin_sqrtf(nlgr1 * nlgr1);
                    const float nl2g = 1 + tau * __builtin_sqrtf(nlgr2 * nlgr2);
                    const float pn = (ns.pq[s][t] + tau * nlgr1) / nl1g;
                    const float qn = (ns.pq[kNlSlots + s][t] + tau * nlgr2) / nl2g;
                    if (in) {
                        ns.pq[s][t] = pn;
                        ns.pq[kNlSlots + s][t] = qn;
                    }
                }
            }
        }
        __syncthreads();
        float term = 0.0f, t1 = u1, t2 = u2;
        if (act) {
            float dp = 0.0f, dq = 0.0f;
            if (NL) {                                               // non_local_divergence, src/aux_energy_model.cpp:178
#pragma unroll
                for (int s = 0; s < kNlSlots; ++s) {
                    int dk, dl;
                    slot_offset<kNlR>(s, dk, dl);
                    const bool in = (nlmask >> s) & 1u;
                    const int qq = in ? t + dk * pw + dl : t;
                    const int rp = kNlSlots - 1 - s;
                    const float a1 = dp + w_[s] * (ns.pq[s][t] - ns.pq[rp][qq]);
                    const float a2 = dq + w_[s] * (ns.pq[kNlSlots + s][t] - ns.pq[kNlSlots + rp][qq]);
                    dp = in ? a1 : dp;
                    dq = in ? a2 : dq;
                }
            } else {
                dp = -0.0f;                                         // -div of the inert TV regulariser (tvcsad_getP)
                dq = -0.0f;
            }
            const float n1 = u1 - tau * (dp + (u1 - v1) / theta);
            const float n2 = u2 - tau * (dq + (u2 - v2) / theta);
            term = (n1 - u1) * (n1 - u1) + (n2 - u2) * (n2 - u2);
            u1 = n1;
            u2 = n2;
            ub1 = 2 * u1 - t1;
            ub2 = 2 * u2 - t2;
        }
        sm.t[t] = term;
        __syncthreads();
        if (t == 0) sm.err[0] = __float_as_int(ordered_sum(sm, np) / (float)((ej - ij) * (ei - ii)));
        sm.a[t] = ub1;
        sm.b[t] = ub2;
        __syncthreads();
        err_D = __int_as_float(sm.err[0]);
    }
    __syncthreads();

    // ---- eval_*: the frame warped by the new flow, mean window energy ----
    float i1n = 0.0f;
    if (act) {
        const BicubicTaps tp = bicubic_taps((float)k + u1, (float)l + u2, nx, ny);
        i1n = bicubic_eval_any(I1, tp, nx, wave_any(tp.clamped));
    }
    sm.a[t] = u1;
    sm.b[t] = u2;
    sm.x[0][t] = i1n;
    __syncthreads();
    float term = 0.0f;
    if (act) {
        ubase[L_U1 * pe + p] = u1;
        ubase[L_U2 * pe + p] = u2;
        const float dc = (1 / (2 * theta)) * ((u1 - v1) * (u1 - v1) + (u2 - v2) * (u2 - v2));
        float g = 0.0f;
        if (NL) {
#pragma unroll
            for (int s = 0; s < kNlSlots; ++s) {
                int dk, dl;
                slot_offset<kNlR>(s, dk, dl);
                const bool in = (nlmask >> s) & 1u;
                const int qq = in ? t + dk * pw + dl : t;
                const double inc = __builtin_fabs((double)(u1 - sm.a[qq])) * (double)w_[s] +
                                   __builtin_fabs((double)(u2 - sm.b[qq])) * (double)w_[s];
                const float gn = (float)((double)g + inc);
                g = in ? gn : g;
                duals[(size_t)s * pe + p] = ns.pq[s][t];
                duals[(size_t)(kNlSlots + s) * pe + p] = ns.pq[kNlSlots + s][t];
            }
            g /= wt;
        }
        float dt;
        if (DATA == kDataL1) {
            dt = weight ? (float)((double)lambda * __builtin_fabs((double)(i1n - i0)) * (double)gwy * (double)gwx)
                        : lambda * __builtin_fabsf(i1n - i0);
        } else {
            dt = 0.0f;
#pragma unroll
            for (int s = 0; s < kCsSlots; ++s) {
                int dk, dl;
                slot_offset<kCsR>(s, dk, dl);
                const bool in = (csmask >> s) & 1ull;
                const int pos = in ? p + dk * nx + dl : p;
                const int qq = in ? t + dk * pw + dl : t;
                const float dn = dt + __builtin_fabsf(i0 - I0[pos] - i1n + sm.x[0][qq]);
                dt = in ? dn : dt;
            }
            dt *= weight ? lambda * gwy * gwx : lambda;
        }
        term = dc + dt + g;
    }
    __syncthreads();                                                 // every read of sm.x[0] is done
    sm.t[t] = term;
    __syncthreads();
    float ener = 0.0f;
    if (t == 0) ener = (float)((double)ordered_sum(sm, np) / (np * 1.0));
    return ener;
}

}  // namespace faldoi
