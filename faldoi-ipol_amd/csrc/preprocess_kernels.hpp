// preprocess_kernels.hpp -- the image preparation global_faldoi's main performs before the
// solver (src/global_faldoi.cpp:2049-2068), on the device: gray conversion, the three-image
// normalisation and the separable Gaussian pre-smoothing.  Bit-identical to the CPU path: the
// same operations in the same order and precision (fp64 where C promotes to double, fp32 else).
#pragma once
#include "faldoi_device.hpp"

#pragma clang fp contract(off)

namespace faldoi {

constexpr int kMaxGaussTaps = 16;
struct GaussTaps {
    float B[kMaxGaussTaps];     // normalised half-kernel, B[0] centre (src/utils.cpp:529-552, host-computed)
    int size;
};

// src/global_faldoi.cpp:1820-1828: .299*R + .587*G + .114*B in double, one rounding to float.
// `rgb` is planar (R plane, G plane, B plane), the layout iio's "split" reader returns.
__global__ void rgb2gray_kernel(const float *__restrict__ rgb, float *__restrict__ out, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = (float)(.299 * (double)rgb[i] + .587 * (double)rgb[n + i] + .114 * (double)rgb[2 * n + i]);
}

// Per-block min/max of up to three planes (src/utils.cpp:679 getminmax; min/max are
// order-independent, so any reduction tree returns the reference's values for NaN-free input).
// part: [gridDim.x][6] = {min0,max0,min1,max1,min2,max2}
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                             const float *__restrict__ c, int n, float *__restrict__ part)
{
    __shared__ float s[6][256];
    const float *src[3] = {a, b, c};
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = src[k][0];
        hi[k] = src[k][0];
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const float v = src[k][i];
            lo[k] = v < lo[k] ? v : lo[k];
            hi[k] = v > hi[k] ? v : hi[k];
        }
        s[2 * k][threadIdx.x] = lo[k];
        s[2 * k + 1][threadIdx.x] = hi[k];
    }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float l2 = s[2 * k][threadIdx.x + st], h2 = s[2 * k + 1][threadIdx.x + st];
                if (l2 < s[2 * k][threadIdx.x]) s[2 * k][threadIdx.x] = l2;
                if (h2 > s[2 * k + 1][threadIdx.x]) s[2 * k + 1][threadIdx.x] = h2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void minmax_final_kernel(const float *__restrict__ part, int nblocks, float *__restrict__ out6)
{
    __shared__ float s[6][256];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const bool is_max = k & 1;
        float v = part[k];
        for (int b = threadIdx.x; b < nblocks; b += 256) {
            const float w = part[b * 6 + k];
            v = is_max ? (w > v ? w : v) : (w < v ? w : v);
        }
        s[k][threadIdx.x] = v;
    }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float a = s[k][threadIdx.x], w = s[k][threadIdx.x + st];
                s[k][threadIdx.x] = (k & 1) ? (w > a ? w : a) : (w < a ? w : a);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) out6[threadIdx.x] = s[threadIdx.x][0];
}

// src/utils.cpp:743-781 image_normalization_3, in place on the three planes.
// mm = {min(A),max(A),min(B),max(B),min(C),max(C)} for the planes bound as the call site binds them
// (src/global_faldoi.cpp:2065): A = i0 ("I1"), B = i1 ("I2"), C = i_1 ("I0").  The value used as
// "min" is the LARGER of min(B) and min(min(C), min(A)) -- reference behaviour (:763), reproduced.
__global__ void normalize3_kernel(float *__restrict__ A, float *__restrict__ B, float *__restrict__ C, int n,
                                  const float *__restrict__ mm)
{
    const float min1 = mm[0], max1 = mm[1], min2 = mm[2], max2 = mm[3], min0 = mm[4], max0 = mm[5];
    const float max01 = (max0 > max1) ? max0 : max1;
    const float mx = (max2 > max01) ? max2 : max01;
    const float min01 = (min0 < min1) ? min0 : min1;
    const float mn = (min2 > min01) ? min2 : min01;
    const float den = mx - mn;
    if (!(den > 0)) return;                              // reference copies the images unchanged
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        A[i] = (A[i] - mn) / den;
        B[i] = (B[i] - mn) / den;
        C[i] = (C[i] - mn) / den;
    }
}

// image_normalization (src/utils.cpp:703-734), two images: the usual min of minima.
__global__ void normalize2_kernel(float *__restrict__ A, float *__restrict__ B, int n, const float *__restrict__ mm)
{
    const float min0 = mm[0], max0 = mm[1], min1 = mm[2], max1 = mm[3];
    const float mx = (max0 > max1) ? max0 : max1;
    const float mn = (min0 < min1) ? min0 : min1;
    const float den = mx - mn;
    if (!(den > 0)) return;                              // reference copies the images unchanged
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        A[i] = (A[i] - mn) / den;
        B[i] = (B[i] - mn) / den;
    }
}

// Index held by the reference's padded line buffer at virtual position i (src/utils.cpp:566-571,
// 600-605): left/top mirror WITHOUT repeating the edge sample, right/bottom mirror WITH it.
__device__ __forceinline__ int reflect_index(int i, int n)
{
    return i < 0 ? -i : (i >= n ? 2 * n - i - 1 : i);
}

// One separable pass of src/utils.cpp:521-630 (stride 1 = rows, stride nx = columns):
// sum = B0*c; sum += B[j]*(left_j + right_j), j ascending, all fp32.
template <bool COLUMNS>
__global__ void gaussian_pass_kernel(const float *__restrict__ in, float *__restrict__ out, int nx, int ny, GaussTaps t)
{
    const int n = nx * ny;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const int y = p / nx, x = p - y * nx;
        const int len = COLUMNS ? ny : nx, pos = COLUMNS ? y : x;
        const int stride = COLUMNS ? nx : 1;
        const float *line = COLUMNS ? in + x : in + y * nx;
        float sum = t.B[0] * line[pos * stride];
        for (int j = 1; j < t.size; ++j) {
            const float a = line[reflect_index(pos - j, len) * stride];
            const float b = line[reflect_index(pos + j, len) * stride];
            sum += t.B[j] * (a + b);
        }
        out[p] = sum;
    }
}

// src/zoom.c:59-73 / :96-105: bicubic samples of `in` (nx x ny) at (j / fx, i / fy) for an nxx x nyy
// output; taps are clamped to the image (border_out = false), never zeroed.
__global__ void zoom_resample_kernel(const float *__restrict__ in, float *__restrict__ out, int nx, int ny, int nxx,
                                     int nyy, float fx, float fy)
{
    const int n = nxx * nyy;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        const int i1 = p / nxx, j1 = p - i1 * nxx;
        const float i2 = (float)i1 / fy;
        const float j2 = (float)j1 / fx;
        const BicubicTaps t = bicubic_taps(j2, i2, nx, ny);
        out[p] = bicubic_eval(in, t, nx);
    }
}

}  // namespace faldoi
