// Stem (7x7/s2/p0 conv 3->64 + folded BN + ReLU, NCHW crop in, NHWC out) and the
// 3x3/s2/p1 max-pool.  reference lib/models/modules.py:70-74,138-141.
//
// The stem is 147 MMAC per 255^2 crop (0.5 % of the frame) with K = 147: too thin for
// the MFMA tile path, so it is a direct convolution on the vector ALUs.  A workgroup owns
// an 8x16 block of output pixels: the 21x37x3 input patch and the whole 147x64 filter bank
// are staged once in LDS; each of the four wavefronts produces 16 of the 64 output
// channels for two pixels per lane, so a tap costs two patch reads plus four broadcast
// ds_read_b128 of the wave's 16 filter values per 32 FMAs.  (Filter taps through the
// scalar cache looked cheaper but serialise: SMEM returns out of order, so every use
// drains lgkmcnt to 0 together with the LDS reads.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ST = 8;                 // output tile rows
constexpr int STW = 16;               // output tile cols (two pixels per lane)
constexpr int PT = 2 * ST + 5;        // 21 patch rows for stride 2, 7 taps
constexpr int PTW = 2 * STW + 5;      // 37 patch cols
constexpr int PTP = PTW + 2;          // padded LDS row

__global__ __launch_bounds__(256) void stem_conv7_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    float *__restrict__ y, int H, int W, int OH, int OW)
{
    __shared__ __attribute__((aligned(16))) float wl[147 * 64];
    __shared__ float patch[3 * PT * PTP];
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * ST, ox0 = blockIdx.x * STW;
    const int iy0 = oy0 * 2, ix0 = ox0 * 2;
    const float *xn = x + (long)n * 3 * H * W;
    // all global loads of the prologue are issued before the first LDS store: a plain
    // load->store loop serialises ~10 dependent HBM round trips per thread (measured 20 us)
    constexpr int NW = (147 * 16 + 255) / 256, NP = (3 * PT * PTW + 255) / 256;
    f32x4 wreg[NW];
    float preg[NP];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int i = threadIdx.x + k * 256;
        wreg[k] = i < 147 * 16 ? ((const f32x4 *)w)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int i = threadIdx.x + k * 256;
        const int ci = i / (PT * PTW), r = i - ci * PT * PTW;
        const int py = r / PTW, px = r - py * PTW;
        const int iy = iy0 + py, ix = ix0 + px;
        preg[k] = (i < 3 * PT * PTW && iy < H && ix < W) ? xn[((long)ci * H + iy) * W + ix] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i < 147 * 16) ((f32x4 *)wl)[i] = wreg[k];
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int i = threadIdx.x + k * 256;
        const int ci = i / (PT * PTW), r = i - ci * PT * PTW;
        const int py = r / PTW, px = r - py * PTW;
        if (i < 3 * PT * PTW) patch[(ci * PT + py) * PTP + px] = preg[k];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int cg = threadIdx.x >> 6;
    const int py = lane >> 3, px = lane & 7;          // pixels (py, px) and (py, px + 8)
    float acc0[16], acc1[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { acc0[c] = 0.f; acc1[c] = 0.f; }
    for (int ci = 0; ci < 3; ++ci) {
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            const float *prow = patch + (ci * PT + 2 * py + kh) * PTP + 2 * px;
            const float *wrow = wl + ((ci * 7 + kh) * 7) * 64 + cg * 16;
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const float x0 = prow[kw], x1 = prow[kw + 16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 wv = *(const f32x4 *)(wrow + kw * 64 + q * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0[q * 4 + e] = fmaf(x0, wv[e], acc0[q * 4 + e]);
                        acc1[q * 4 + e] = fmaf(x1, wv[e], acc1[q * 4 + e]);
                    }
                }
            }
        }
    }
    const int oy = oy0 + py;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int ox = ox0 + px + 8 * half;
        if (oy < OH && ox < OW) {
            float *yo = y + (((long)n * OH + oy) * OW + ox) * 64 + cg * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = fmaxf((half ? acc1[q * 4 + e] : acc0[q * 4 + e]) + bias[cg * 16 + q * 4 + e], 0.f);
                *(f32x4 *)(yo + q * 4) = v;
            }
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
    if ((uintptr_t)w % 16) return USOT_EINVAL;
    dim3 grid(usot_cdiv(OW, STW), usot_cdiv(OH, ST), N);
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
