// Micro-benchmark (tuning aid, not product): issue cost of the VALU instructions the TV-L1
// kernel is made of, on gfx950, at 1/2/4/8 waves per SIMD.  Prints cycles per wave-instruction
// per SIMD (wall cycles * SIMDs busy / instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)

typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 2000
#define UNR 16

template <int OP>
__global__ void k(float *out, float seed, double dseed)
{
    float a[UNR];
    f2 p[UNR];
    double d[UNR];
#pragma unroll
    for (int i = 0; i < UNR; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1}; d[i] = dseed + i; }
    const float b = seed * 0.5f, c = seed * 0.25f;
    const bool sel = (threadIdx.x & 7) != 3;
    const unsigned msk = (threadIdx.x & 7) != 3 ? 0xffffffffu : 0x7fffff00u, sgn = (threadIdx.x & 15) == 5 ? 0x80000000u : 0u;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], b, c);
            if (OP == 1) a[i] = a[i] * b;
            if (OP == 2) a[i] = a[i] + b;
            if (OP == 3) p[i] = __builtin_elementwise_fma(p[i], f2{b, b}, f2{c, c});
            if (OP == 4) p[i] = p[i] * f2{b, c};
            if (OP == 5) p[i] = p[i] + f2{b, c};
            if (OP == 6) d[i] = __builtin_fma(d[i], (double)b, (double)c);
            if (OP == 7) d[i] = d[i] * (double)b;
            if (OP == 8) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == 9) a[i] = __builtin_amdgcn_sqrtf(a[i]);
            if (OP == 10) d[i] = __builtin_amdgcn_rcp(d[i]);
            if (OP == 11) a[i] = (float)((double)a[i] * dseed);          // cvt, mul_f64, cvt
            if (OP == 12) a[i] = a[i] / b;                                // IEEE fp32 division
            if (OP == 13) a[i] = a[i] > b ? c : a[i];                     // cmp + cndmask
            if (OP == 14) a[i] = __builtin_sqrtf(a[i]);                   // IEEE sqrt
            if (OP == 15) a[i] = __builtin_amdgcn_div_fixup(a[i], b, c);
            if (OP == 16) a[i] = __shfl_down(a[i], 1);
            if (OP == 17) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x130, 0xf, 0xf, false));
            if (OP == 18) a[i] = __builtin_fmaxf(a[i], b);
            if (OP == 19) { d[i] = (double)a[i] + d[i]; }                // cvt f32->f64 + add_f64 (2 ops)
            if (OP == 20) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x111, 0xf, 0xf, false));   // row_shr:1
            if (OP == 21) a[i] = a[i] - __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x130, 0xf, 0xf, true)); // v_sub_f32_dpp wave_shl:1
            if (OP == 22) a[i] = a[i] - __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x101, 0xf, 0xf, true)); // v_sub_f32_dpp row_shl:1
            if (OP == 23) a[i] = (float)__builtin_amdgcn_frexp_expf(a[i]) + b;  // frexp_exp + cvt + add (3 ops)
            if (OP == 24) a[i] = (float)((double)a[i]);                        // cvt + cvt (2 ops)
            if (OP == 26) a[i] = __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, a[i]), 0x8000) ? a[i] : b;       // ds_swizzle
            if (OP == 27) a[i] = sel ? a[i] + b : a[i];                        // add + cndmask on a loop-invariant lane mask (2 ops)
            if (OP == 28) a[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a[i] + b) & msk);   // add + v_and_b32 (2 ops)
            if (OP == 29) a[i] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, a[i] + b) & msk) | sgn);   // add + v_and_or_b32 (2 ops)
            if (OP == 30) a[i] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, a[i]) << 1) - 1u);         // v_lshl_add_u32
            if (OP == 31) { const unsigned x = __builtin_bit_cast(unsigned, a[i]); a[i] = __builtin_bit_cast(float, x < msk ? x : msk); }   // v_min_u32
            if (OP == 32) a[i] = __builtin_amdgcn_ballot_w64(a[i] > b) ? a[i] + c : a[i] + b;   // cmp + ballot + scalar select of the addend + add
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNR; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, float instr_per_op, float *out)
{
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps;     // waves spread over 4 SIMDs
        const int blocks_per_cu = (64 * 4 * wps) / threads;
        const int grid = 256 * blocks_per_cu;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, out, 1.0001f, 1.0001);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, out, 1.0001f, 1.0001);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)wps * ITERS * UNR * instr_per_op;
        printf("%-28s waves/SIMD=%d  %.3f ms  %.2f ns per wave-op per SIMD (%.2f cyc @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}

int main()
{
    float *out; hipMalloc(&out, sizeof(float) * 1024 * 2048);
    run<0>("v_fma_f32", 1, out);
    run<1>("v_mul_f32", 1, out);
    run<2>("v_add_f32", 1, out);
    run<3>("v_pk_fma_f32", 1, out);
    run<4>("v_pk_mul_f32", 1, out);
    run<5>("v_pk_add_f32", 1, out);
    run<6>("v_fma_f64", 1, out);
    run<7>("v_mul_f64", 1, out);
    run<8>("v_rcp_f32", 1, out);
    run<9>("v_sqrt_f32", 1, out);
    run<10>("v_rcp_f64", 1, out);
    run<11>("cvt+mul_f64+cvt (3 ops)", 1, out);
    run<12>("IEEE fp32 div (per div)", 1, out);
    run<13>("cmp+cndmask (2 ops)", 1, out);
    run<14>("IEEE sqrtf (per sqrt)", 1, out);
    run<15>("v_div_fixup_f32", 1, out);
    run<16>("__shfl_down (bpermute)", 1, out);
    run<17>("dpp wave_shl mov", 1, out);
    run<18>("v_max_f32", 1, out);
    run<19>("cvt_f64_f32 + add_f64 (2 ops)", 1, out);
    run<20>("dpp row_shr:1 mov", 1, out);
    run<21>("v_sub_f32_dpp wave_shl:1", 1, out);
    run<22>("v_sub_f32_dpp row_shl:1", 1, out);
    run<23>("frexp_exp+cvt+add (3 ops)", 1, out);
    run<24>("cvt f32->f64->f32 (2 ops)", 1, out);
    run<27>("add + cndmask, invariant mask (2 ops)", 1, out);
    run<28>("add + v_and_b32 (2 ops)", 1, out);
    run<29>("add + v_and_or_b32 (2 ops)", 1, out);
    run<30>("v_lshl_add_u32", 1, out);
    run<31>("v_min_u32", 1, out);
    run<32>("cmp + ballot + s_cselect + add", 1, out);
    return 0;
}
