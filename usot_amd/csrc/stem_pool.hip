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
    float *__restrict__ y, int H, int W, int OH, int OW, float mu0, float mu1, float mu2)
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
        // x - mu[ci] (exact for pad 0: every tap of a valid output reads a real pixel; the caller's bias carries sum(w) * mu)
        preg[k] = (i < 3 * PT * PTW && iy < H && ix < W) ? xn[((long)ci * H + iy) * W + ix] - (ci == 0 ? mu0 : ci == 1 ? mu1 : mu2) : 0.f;
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


// ---------------------------------------------------------------------------------------
// Fused fp32 stem + max-pool on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate): the tracked
// frame never needs the 125x125x64 stem map itself, only its 3x3/s2 pooled image (modules.py:138-141).
// k axis as in the low-precision kernel: 24 rows (ci*7 + kh; rows 21-23 zero) x 8 taps (kw 0-6 + one zero
// tap) = 48 MFMA k-steps of 4: step s, lane quad q -> row s/2, tap (s&1)*4 + q.  Wavefront w owns channel
// block w (16 channels) for every pixel of the tile, so its 48 A operands (filters) live in registers and a
// B operand is one ds_read_b32 of the staged crop patch per MFMA.  A workgroup = 4x4 pooled pixels = 9x9
// stem pixels (halo recomputed) = a 23x23 patch; stem outputs meet in LDS for the pool.
constexpr int FP_P = 4;                              // pooled tile edge
constexpr int FP_S = 2 * FP_P + 1;                   // 9 stem pixels per edge
constexpr int FP_NPIX = FP_S * FP_S;                 // 81
constexpr int FP_NBLK = (FP_NPIX + 15) / 16;         // 6 MFMA pixel blocks
constexpr int FP_I = 2 * FP_S + 5;                   // 23 patch rows / columns
constexpr int FP_IP = FP_I + 1;                      // padded patch row (the zero tap reads one past)
constexpr int FP_OP = 64 + 4;                        // stem-tile LDS row pitch (floats)

__global__ __launch_bounds__(256) void stem_pool_f32_kernel(
    const float *__restrict__ x, const float *__restrict__ wfrag, const float *__restrict__ bias,
    float *__restrict__ y, int H, int W, int OH, int OW, int PH, int PW, const int *__restrict__ xptr,
    float mu0, float mu1, float mu2)
{
    // xptr != NULL: the crop's address comes from DEVICE memory (two ints: low / high half), written earlier in the
    // same graph from the host's control block — a captured frame can then read ANY resident crop, no copy into a
    // baked input buffer (a null address keeps the baked `x`)
    if (xptr) {
        const unsigned long long a = ((unsigned long long)(unsigned)xptr[1] << 32) | (unsigned)xptr[0];
        if (a) x = (const float *)a;
    }
    __shared__ float patch[3 * FP_I * FP_IP + 8];
    __shared__ __attribute__((aligned(16))) float stile[FP_NBLK * 16 * FP_OP];
    const int n = blockIdx.z;
    const int py0 = blockIdx.y * FP_P, px0 = blockIdx.x * FP_P;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;
    const int iy0 = 2 * sy0, ix0 = 2 * sx0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, quad = lane >> 4;

    float wf[48];                                    // this wave's channel block, [step][lane] in global
#pragma unroll
    for (int st = 0; st < 48; ++st) wf[st] = wfrag[(wave * 48 + st) * 64 + lane];

    const float *xn = x + (long)n * 3 * H * W;
    constexpr int NPL = (3 * FP_I * FP_IP + 8 + 255) / 256;
    float pv[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int i = tid + q * 256;
        const int ci = i / (FP_I * FP_IP), r = i - ci * FP_I * FP_IP;
        const int py = r / FP_IP, px = r - py * FP_IP;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        // x - mu[ci]: raw crops are 0..255 around ~100 (track_utils.py:24-27 feeds them unnormalised); without the offset
        // every partial sum of the 147-tap chain rides on 100 * sum(w).  Exact for pad 0 (only real pixels reach a valid
        // output; out-of-image patch entries feed halo stem pixels the pool never reads); the bias carries sum(w) * mu.
        if (ci < 3 && px < FP_I && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = xn[((long)ci * H + iy) * W + ix] - (ci == 0 ? mu0 : ci == 1 ? mu1 : mu2);
        pv[q] = v;
    }
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int i = tid + q * 256;
        if (i < 3 * FP_I * FP_IP + 8) patch[i] = pv[q];
    }
    __syncthreads();

    const f32x4 bv = *(const f32x4 *)(bias + wave * 16 + quad * 4);
#pragma unroll 1
    for (int blk = 0; blk < FP_NBLK; ++blk) {
        int pi = blk * 16 + l15;
        if (pi > FP_NPIX - 1) pi = FP_NPIX - 1;
        const int sy = pi / FP_S, sx = pi - sy * FP_S;
        const float *pb = patch + (2 * sy) * FP_IP + 2 * sx + quad;      // + row/tap offset of the step
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
        for (int st = 0; st < 48; ++st) {
            const int row = st / 2 > 20 ? 20 : st / 2;                   // zero-filter rows: any valid address
            const int off = ((row / 7) * FP_I + row % 7) * FP_IP + (st & 1) * 4;
            if (st & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[st], pb[off], acc2, 0, 0, 0);
            else        acc  = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[st], pb[off], acc, 0, 0, 0);
        }
        f32x4 v = acc + acc2 + bv;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        *(f32x4 *)(stile + (blk * 16 + l15) * FP_OP + wave * 16 + quad * 4) = v;
    }
    __syncthreads();

    // 3x3 / stride 2 / pad 1 max-pool: 16 pooled pixels x 16 channel quads
    const int pp = tid >> 4, c4 = tid & 15;
    const int ppy = pp / FP_P, ppx = pp - ppy * FP_P;
    const int py = py0 + ppy, px = px0 + ppx;
    if (py >= PH || px >= PW) return;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int ly = 2 * ppy + dy, gy = sy0 + ly;
        if ((unsigned)gy >= (unsigned)OH) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int lx = 2 * ppx + dx, gx = sx0 + lx;
            if ((unsigned)gx >= (unsigned)OW) continue;
            const f32x4 v = *(const f32x4 *)(stile + (ly * FP_S + lx) * FP_OP + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
        }
    }
    *(f32x4 *)(y + ((((long)n * PH + py) * PW + px) * 64 + c4 * 4)) = m;
}

}  // namespace

extern "C" int usot_stem_conv_f32(void *stream, const float *x, const float *w, const float *bias,
                                  float *y, int N, int H, int W, int OH, int OW)
{
    return usot_stem_conv_mu_f32(stream, x, w, bias, y, N, H, W, OH, OW, 0.f, 0.f, 0.f);
}

extern "C" int usot_stem_conv_mu_f32(void *stream, const float *x, const float *w, const float *bias,
                                     float *y, int N, int H, int W, int OH, int OW, float mu0, float mu1, float mu2)
{
    if (!x || !w || !bias || !y || N <= 0 || H < 7 || W < 7) return USOT_EINVAL;
    if (OH != (H - 7) / 2 + 1 || OW != (W - 7) / 2 + 1) return USOT_EINVAL;
    if ((uintptr_t)y % 16) return USOT_EINVAL;
    if ((uintptr_t)w % 16) return USOT_EINVAL;
    dim3 grid(usot_cdiv(OW, STW), usot_cdiv(OH, ST), N);
    hipLaunchKernelGGL(stem_conv7_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, H, W, OH, OW, mu0, mu1, mu2);
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

/* Fused fp32 stem + max-pool (frame plans: the stem map itself is not needed).  wfrag = the BN-folded
 * filter bank as MFMA A operands [4 channel blocks][48 k-steps][64 lanes] (usot_amd/engine.py:
 * pack_stem_f32), bias fp32[64], y NHWC [N][PH][PW][64]. */

extern "C" int usot_stem_pool_f32(void *stream, const float *x, const float *wfrag, const float *bias, float *y,
                                  int N, int H, int W, int OH, int OW, int PH, int PW)
{
    return usot_stem_pool_mu_f32(stream, x, wfrag, bias, y, N, H, W, OH, OW, PH, PW, nullptr, 0.f, 0.f, 0.f);
}

extern "C" int usot_stem_pool_ind_f32(void *stream, const float *x, const float *wfrag, const float *bias, float *y,
                                      int N, int H, int W, int OH, int OW, int PH, int PW, const int32_t *xptr_dev)
{
    return usot_stem_pool_mu_f32(stream, x, wfrag, bias, y, N, H, W, OH, OW, PH, PW, xptr_dev, 0.f, 0.f, 0.f);
}

extern "C" int usot_stem_pool_mu_f32(void *stream, const float *x, const float *wfrag, const float *bias, float *y,
                                     int N, int H, int W, int OH, int OW, int PH, int PW, const int32_t *xptr_dev,
                                     float mu0, float mu1, float mu2)
{
    if (!x || !wfrag || !bias || !y || N <= 0 || N > 65535 || H < 7 || W < 7) return USOT_EINVAL;
    if (OH != (H - 7) / 2 + 1 || OW != (W - 7) / 2 + 1) return USOT_EINVAL;
    if (PH != (OH + 2 - 3) / 2 + 1 || PW != (OW + 2 - 3) / 2 + 1) return USOT_EINVAL;
    if (((uintptr_t)y % 16) || ((uintptr_t)bias % 16)) return USOT_EINVAL;
    dim3 grid(usot_cdiv(PW, FP_P), usot_cdiv(PH, FP_P), N);
    hipLaunchKernelGGL(stem_pool_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, wfrag, bias, y, H, W, OH, OW, PH, PW,
                       (const int *)xptr_dev, mu0, mu1, mu2);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}
