// Probe: do the IEEE exception status bits of TRAPSTS accumulate on gfx950 with traps disabled?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#define GETREG_TRAPSTS() __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 3)
#define GETREG_MODE() __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 1)
__global__ void probe(const float *a, const float *b, float *out, unsigned *flags, double one)
{
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const int w = t >> 6;
    unsigned f0 = GETREG_TRAPSTS();
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0" ::: "memory");
    unsigned f1 = GETREG_TRAPSTS();
    float x = a[t], y = b[t];
    asm volatile("" : "+v"(x), "+v"(y));
    float p = x * y;                                   // fp32 product: may underflow
    asm volatile("" : "+v"(p));
    int dep = __builtin_amdgcn_readfirstlane(__float_as_int(p));
    asm volatile("" : "+s"(dep));
    unsigned f2 = GETREG_TRAPSTS();
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0" ::: "memory");
    double yd = (double)y * one; asm volatile("" : "+v"(yd));
    double q = (double)x * yd;
    float c = (float)q;                                // conversion: may underflow
    asm volatile("" : "+v"(c));
    dep = __builtin_amdgcn_readfirstlane(__float_as_int(c));
    asm volatile("" : "+s"(dep));
    unsigned f3 = GETREG_TRAPSTS();
    out[2 * t] = p; out[2 * t + 1] = c;
    if ((t & 63) == 0) { flags[5 * w] = f0; flags[5 * w + 1] = f1; flags[5 * w + 2] = f2; flags[5 * w + 3] = f3; flags[5 * w + 4] = GETREG_MODE(); }
}
int main()
{
    const int W = 6, N = 64 * W;
    float ha[N], hb[N];
    for (int i = 0; i < N; ++i) { ha[i] = 1.5f; hb[i] = 2.25f; }                       // wave 0: exact products
    for (int i = 64; i < 128; ++i) { ha[i] = 1.1f; hb[i] = 1.3f; }                     // wave 1: inexact, normal
    ha[128 + 5] = 1e-30f; hb[128 + 5] = 1.1e-10f;                                      // wave 2: one lane tiny inexact (subnormal result)
    ha[192 + 7] = ldexpf(1.0f, -100); hb[192 + 7] = ldexpf(1.0f, -30);                 // wave 3: one lane tiny EXACT (2^-130)
    ha[256 + 9] = ldexpf(3.0f, -149); hb[256 + 9] = 0.5f;                              // wave 4: denormal input, tie result
    ha[320 + 1] = 1e-30f; hb[320 + 1] = 1e-30f;                                        // wave 5: underflow to zero
    float *da, *db, *dout; unsigned *df;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, 2 * sizeof ha); hipMalloc(&df, 5 * W * 4);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    probe<<<1, N>>>(da, db, dout, df, 1.0);
    unsigned hf[5 * W]; float ho[2 * N];
    if (hipMemcpy(hf, df, sizeof hf, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    const char *what[W] = { "exact", "inexact normal", "tiny inexact", "tiny exact", "denormal in, tie", "underflow to 0" };
    for (int w = 0; w < W; ++w)
        printf("wave %d (%-18s) entry %08x cleared %08x after fp32 mul EXCP=%03x after f64->f32 cvt EXCP=%03x MODE=%08x\n", w, what[w], hf[5 * w], hf[5 * w + 1],
               hf[5 * w + 2] & 0x1ff, hf[5 * w + 3] & 0x1ff, hf[5 * w + 4]);
    printf("lane results: w2 %a %a | w3 %a %a | w4 %a %a | w5 %a %a\n", ho[2 * 133], ho[2 * 133 + 1], ho[2 * 199], ho[2 * 199 + 1], ho[2 * 265], ho[2 * 265 + 1], ho[2 * 321], ho[2 * 321 + 1]);
    return 0;
}
