// Stem (7x7/s2/p0 conv 3->64 + folded BN + ReLU, NCHW crop in, NHWC out) and the
// 3x3/s2/p1 max-pool.  reference lib/models/modules.py:70-74,138-141.
//
// The stem is 147 MMAC per 255^2 crop (0.5 % of the frame) with K = 147: too thin for
// the MFMA tile path, so it is a direct convolution on the vector ALUs.  A workgroup owns
// an 8x8 block of output pixels: the 21x21x3 input patch is staged once in LDS; each of
// the four wavefronts produces 16 of the 64 output channels, so its 16 filter taps per
// (ci,kh,kw) are wave-uniform and arrive through the scalar cache (s_load), leaving one
// LDS read per 16 FMAs on the vector side.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ST = 8;                 // output tile edge
constexpr int PT = 2 * ST + 5;        // 21: input patch edge for stride 2, 7 taps
constexpr int PTP = PT + 2;           // padded LDS row

__global__ __launch_bounds__(256) void stem_conv7_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ y, int H, int W, int OH, int OW)
{
    __shared__ float patch[3 * PT * PTP];
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * ST, ox0 = blockIdx.x * ST;
    const int iy0 = oy0 * 2, ix0 = ox0 * 2;
    const float *xn = x + (long)n * 3 * H * W;
    for (int i = threadIdx.x; i < 3 * PT * PT; i += 256) {
        const int ci = i / (PT * PT), r = i - ci * PT * PT;
        const int py = r / PT, px = r - py * PT;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if (iy < H && ix < W) v = xn[((long)ci * H + iy) * W + ix];
        patch[(ci * PT + py) * PTP + px] = v;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    const int py = lane >> 3, px = lane & 7;
    const float *wq = w + cg * 16;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (int ci = 0; ci < 3; ++ci) {
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            const float *prow = patch + (ci * PT + 2 * py + kh) * PTP + 2 * px;
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const float xv = prow[kw];
                const float *wt = wq + ((ci * 7 + kh) * 7 + kw) * 64;
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(xv, wt[c], acc[c]);
            }
        }
    }
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy < OH && ox < OW) {
        float *yo = y + (((long)n * OH + oy) * OW + ox) * 64 + cg * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[q * 4 + e] + bias[cg * 16 + q * 4 + e], 0.f);
            *(f32x4 *)(yo + q * 4) = v;
        }
    }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(
    const float *__restrict__ x, float *__restrict__ y, int N, int H, int W, int C4, int OH, int OW)
{
    const long total = (long)N * OH * OW * C4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C4);
        long r = idx / C4;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *(const f32x4 *)(x + ((((long)n * H + iy) * W + ix) * C4 + c) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        *(f32x4 *)(y + idx * 4) = m;
    }
}

}  // namespace

extern "C" int usot_stem_conv_f32(void *stream, const float *x, const float *w, const float *bias,
                                  float *y, int N, int H, int W, int OH, int OW)
{
    if (!x || !w || !bias || !y || N <= 0 || H < 7 || W < 7) return USOT_EINVAL;
    if (OH != (H - 7) / 2 + 1 || OW != (W - 7) / 2 + 1) return USOT_EINVAL;
    if ((uintptr_t)y % 16) return USOT_EINVAL;
    dim3 grid(usot_cdiv(OW, ST), usot_cdiv(OH, ST), N);
    hipLaunchKernelGGL(stem_conv7_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, H, W, OH, OW);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_maxpool3x3s2_f32(void *stream, const float *x, float *y,
                                     int N, int H, int W, int C, int OH, int OW)
{
    if (!x || !y || N <= 0 || (C & 3)) return USOT_EINVAL;
    if (OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1) return USOT_EINVAL;
    if (((uintptr_t)x % 16) || ((uintptr_t)y % 16)) return USOT_EINVAL;
    const long total = (long)N * OH * OW * (C / 4);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       x, y, N, H, W, C / 4, OH, OW);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}
