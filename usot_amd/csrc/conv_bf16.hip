// bf16 implicit-GEMM convolution on v_mfma_f32_16x16x32_bf16 (gfx950's 2x-K bf16 MFMA),
// fp32 accumulate, bf16 NHWC in/out.  Used for the batched backbone (BASELINE config 3:
// 64 crops, the MFMA-roofline run); the batch-1 tracking path stays fp32 (1e-4 parity).
//
// Same GEMM orientation as the fp32 kernel (weights = MFMA A operand, activations = B):
// a lane's four accumulator registers are four consecutive output channels of one pixel,
// so bias / residual / ReLU / the bf16 pack are per-lane and the store is 8 bytes.
// A lane feeds one MFMA with 8 consecutive k (16 bytes) of its row: one ds_read_b128 per
// operand per 16x16x32 step, quad q of the wave owning k-chunk q.  k-tile = 64 bf16 (128 B
// per row, padded to 144 B in LDS), register-staged double buffering as in conv_igemm_f32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct ConvB {
    const uint16_t *x, *w, *res;
    const float *bias;
    uint16_t *y;
    int N, H, W, Cin, OH, OW, Cout;
    int KH, KW, stride, pad_h, pad_w, dil_h, dil_w;
    int act;
    int out_f32;       // store the result as fp32 (the neck output that feeds the fp32 heads)
    int M, K, KT, cchunks, MT, NT, P;
};

__device__ __forceinline__ int xcd_remap_b(int b, int total)
{
    const int q = total >> 3, r = total & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ uint32_t f2bf(float f)          // round to nearest even
{
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf2f(uint32_t h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ uint32_t f2h(float f) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float h2f(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
template <bool F16> __device__ __forceinline__ uint32_t pack_lp(float f) { return F16 ? f2h(f) : f2bf(f); }
template <bool F16> __device__ __forceinline__ float unpack_lp(uint32_t h) { return F16 ? h2f(h) : bf2f(h); }

constexpr int BKB = 64;            // k-tile in bf16 elements
constexpr int LDC = 9;             // 16-byte chunks per LDS row (8 data + 1 pad)

template <int BM, int BN, int WM, int WN, bool F16>
__global__ __launch_bounds__(256) void conv_igemm_bf16(const ConvB p)
{
    static_assert(WM * WN == 4, "4 wavefronts");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int XI = (BM + 31) / 32, WI = (BN + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
    u32x4 *sX = smem4;                      // [2][BM][LDC]
    u32x4 *sW = smem4 + 2 * BM * LDC;       // [2][BN][LDC]

    const int tid = threadIdx.x;
    const int tiles = p.MT * p.NT;
    const int b = xcd_remap_b(blockIdx.x, tiles);
    const int bn0 = (b / p.MT) * BN, bm0 = (b % p.MT) * BM;

    const int lr = tid >> 3, kc = tid & 7;
    int x_ih0[XI], x_iw0[XI];
    long x_nb[XI];
    bool x_ok[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int m = bm0 + lr + 32 * i;
        x_ok[i] = (lr + 32 * i < BM) && (m < p.M);
        const int mm = x_ok[i] ? m : 0;
        const int n = mm / p.P, pix = mm - n * p.P;
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        x_ih0[i] = oh * p.stride - p.pad_h;
        x_iw0[i] = ow * p.stride - p.pad_w;
        x_nb[i] = (long)n * p.H * p.W * p.Cin + kc * 8;
    }
    const uint16_t *wp[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int co = bn0 + lr + 32 * i;
        wp[i] = p.w + (long)((lr + 32 * i < BN && co < p.Cout) ? co : 0) * p.K + kc * 8;
    }
    const uint16_t *xp[XI];
    bool xin[XI];
    int cur_tap = 0, cur_cc = 0;
    auto set_tap = [&](int tap) {
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int ih = x_ih0[i] + kh * p.dil_h, iw = x_iw0[i] + kw * p.dil_w;
            xin[i] = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            xp[i] = p.x + x_nb[i] + ((long)ih * p.W + iw) * p.Cin;
        }
    };
    set_tap(0);
    u32x4 xr[XI], wr[WI];
    auto load_tile = [&](bool advance) {
        const int c0 = cur_cc * BKB;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (xin[i]) v = *(const u32x4 *)(xp[i] + c0);
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            wr[i] = *(const u32x4 *)wp[i];
            wp[i] += advance ? BKB : 0;
        }
        if (advance && ++cur_cc == p.cchunks) {
            cur_cc = 0;
            set_tap(++cur_tap);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XI; ++i)
            if (BM % 32 == 0 || lr + 32 * i < BM) sX[(buf * BM + lr + 32 * i) * LDC + kc] = xr[i];
#pragma unroll
        for (int i = 0; i < WI; ++i)
            if (BN % 32 == 0 || lr + 32 * i < BN) sW[(buf * BN + lr + 32 * i) * LDC + kc] = wr[i];
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l15 = lane & 15, quad = lane >> 4;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.KT;
    load_tile(nt > 1);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < nt;
        if (more) load_tile(t + 2 < nt);
        const u32x4 *cX = sX + (cur * BM + wm * TM * 16 + l15) * LDC + quad;
        const u32x4 *cW = sW + (cur * BN + wn * TN * 16 + l15) * LDC + quad;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = cW[i * 16 * LDC + ks * 4];
#pragma unroll
            for (int j = 0; j < TM; ++j) xf[j] = cX[j * 16 * LDC + ks * 4];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    if constexpr (F16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i]),
                                                                           __builtin_bit_cast(f16x8, xf[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i]),
                                                                            __builtin_bit_cast(bf16x8, xf[j]), acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = bm0 + (wm * TM + j) * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int co = bn0 + (wn * TN + i) * 16 + quad * 4;
            if (co >= p.Cout) continue;
            f32x4 v = acc[i][j];
            if (p.bias) v += *(const f32x4 *)(p.bias + co);
            if (p.res) {
                const u32x2 r = *(const u32x2 *)(p.res + (long)m * p.Cout + co);
                v[0] += unpack_lp<F16>(r[0] & 0xffffu); v[1] += unpack_lp<F16>(r[0] >> 16);
                v[2] += unpack_lp<F16>(r[1] & 0xffffu); v[3] += unpack_lp<F16>(r[1] >> 16);
            }
            if (p.act == USOT_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (p.out_f32) {
                *(f32x4 *)((float *)p.y + (long)m * p.Cout + co) = v;
            } else {
                u32x2 o;
                o[0] = pack_lp<F16>(v[0]) | (pack_lp<F16>(v[1]) << 16);
                o[1] = pack_lp<F16>(v[2]) | (pack_lp<F16>(v[3]) << 16);
                *(u32x2 *)(p.y + (long)m * p.Cout + co) = o;
            }
        }
    }
}

struct TileB { int bm, bn; void (*fn)(const ConvB); void (*fn16)(const ConvB); };
#define TB(bm, bn, wm, wn) { bm, bn, conv_igemm_bf16<bm, bn, wm, wn, false>, conv_igemm_bf16<bm, bn, wm, wn, true> }
const TileB kTilesB[] = {
    TB(128, 128, 2, 2),   // 1
    TB(128, 64, 2, 2),    // 2
    TB(64, 128, 2, 2),    // 3
    TB(64, 64, 2, 2),     // 4
    TB(32, 64, 2, 2),     // 5
};
constexpr int kNumTilesB = sizeof(kTilesB) / sizeof(kTilesB[0]);

// fp32 -> bf16 (and back) elementwise, 8 elements per thread
template <bool F16>
__global__ __launch_bounds__(256) void cvt_f32_bf16_kernel(const float *__restrict__ src, uint16_t *__restrict__ dst, long n8)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f32x4 a = ((const f32x4 *)src)[2 * i], b = ((const f32x4 *)src)[2 * i + 1];
        u32x4 o;
        o[0] = pack_lp<F16>(a[0]) | (pack_lp<F16>(a[1]) << 16); o[1] = pack_lp<F16>(a[2]) | (pack_lp<F16>(a[3]) << 16);
        o[2] = pack_lp<F16>(b[0]) | (pack_lp<F16>(b[1]) << 16); o[3] = pack_lp<F16>(b[2]) | (pack_lp<F16>(b[3]) << 16);
        ((u32x4 *)dst)[i] = o;
    }
}

// 3x3/s2/p1 max-pool on bf16 NHWC, 8 channels per thread (bf16 max is exact via float compare)
template <bool F16>
__global__ __launch_bounds__(256) void maxpool3x3s2_bf16_kernel(
    const uint16_t *__restrict__ x, uint16_t *__restrict__ y, int N, int H, int W, int C8, int OH, int OW)
{
    const long total = (long)N * OH * OW * C8;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C8);
        long r = idx / C8;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const u32x4 v = ((const u32x4 *)x)[(((long)n * H + iy) * W + ix) * C8 + c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m[2 * e] = fmaxf(m[2 * e], unpack_lp<F16>(v[e] & 0xffffu));
                    m[2 * e + 1] = fmaxf(m[2 * e + 1], unpack_lp<F16>(v[e] >> 16));
                }
            }
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_lp<F16>(m[2 * e]) | (pack_lp<F16>(m[2 * e + 1]) << 16);
        ((u32x4 *)y)[idx] = o;
    }
}

}  // namespace

extern "C" int usot_conv_bf16_tile_count(void) { return kNumTilesB; }

/* bf16 NHWC conv: x/w/res/y are bf16 (uint16 storage), bias fp32.  Uses the fields N..dil_w,
 * act (NONE or RELU), tile of usot_conv_desc; y dense NHWC [N][OH][OW][Cout]; res same layout;
 * Cin % 64 == 0, Cout % 4 == 0; groups / ksplit / nchw output are fp32-path features. */
extern "C" int usot_conv2d_lp(void *stream, const usot_conv_desc *d, int dtype, int out_f32)
{
    if (dtype != 0 && dtype != 1) return USOT_EINVAL;
    if (!d || !d->x || !d->w || !d->y) return USOT_EINVAL;
    if (d->Cin <= 0 || (d->Cin % BKB) || d->Cout <= 0 || (d->Cout & 3) || d->N <= 0) return USOT_EINVAL;
    if (d->groups > 1 || d->ksplit > 1 || d->y_nchw) return USOT_EINVAL;
    if (d->act != USOT_ACT_NONE && d->act != USOT_ACT_RELU) return USOT_EINVAL;
    const int oh = (d->H + 2 * d->pad_h - d->dil_h * (d->KH - 1) - 1) / d->stride + 1;
    const int ow = (d->W + 2 * d->pad_w - d->dil_w * (d->KW - 1) - 1) / d->stride + 1;
    if (oh != d->OH || ow != d->OW || oh <= 0 || ow <= 0) return USOT_EINVAL;
    if (((uintptr_t)d->x % 16) || ((uintptr_t)d->w % 16) || ((uintptr_t)d->y % (out_f32 ? 16 : 8)) ||
        (d->res && (uintptr_t)d->res % 8) || (d->bias && (uintptr_t)d->bias % 16)) return USOT_EINVAL;
    ConvB p;
    p.x = (const uint16_t *)d->x; p.w = (const uint16_t *)d->w; p.res = (const uint16_t *)d->res;
    p.bias = d->bias; p.y = (uint16_t *)d->y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    p.dil_h = d->dil_h; p.dil_w = d->dil_w; p.act = d->act; p.out_f32 = out_f32;
    p.P = d->OH * d->OW; p.M = d->N * p.P; p.K = d->KH * d->KW * d->Cin;
    p.cchunks = d->Cin / BKB; p.KT = d->KH * d->KW * p.cchunks;
    int tile = d->tile;
    if (tile == 0) {
        const long b128 = (long)((p.M + 127) / 128) * ((d->Cout + 127) / 128);
        tile = (b128 >= 512 && d->Cout >= 128) ? 1 : ((long)((p.M + 63) / 64) * ((d->Cout + 63) / 64) >= 512 ? 4 : 5);
    }
    if (tile < 1 || tile > kNumTilesB) return USOT_EINVAL;
    const TileB &tc = kTilesB[tile - 1];
    p.MT = (p.M + tc.bm - 1) / tc.bm;
    p.NT = (d->Cout + tc.bn - 1) / tc.bn;
    const long blocks = (long)p.MT * p.NT;
    if (blocks <= 0 || blocks > 0x7fffffffL) return USOT_EINVAL;
    const size_t lds = (size_t)2 * (tc.bm + tc.bn) * LDC * 16;
    hipLaunchKernelGGL(dtype ? tc.fn16 : tc.fn, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_conv2d_bf16(void *stream, const usot_conv_desc *d) { return usot_conv2d_lp(stream, d, 0, 0); }

extern "C" int usot_cvt_f32_to_lp(void *stream, const float *src, void *dst, int64_t n, int dtype)
{
    if (!src || !dst || n <= 0 || (n & 7) || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16)) return USOT_EINVAL;
    const long n8 = n / 8;
    const int blocks = (int)((n8 + 255) / 256 > 8192 ? 8192 : (n8 + 255) / 256);
    if (dtype) hipLaunchKernelGGL(cvt_f32_bf16_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (uint16_t *)dst, n8);
    else       hipLaunchKernelGGL(cvt_f32_bf16_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (uint16_t *)dst, n8);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_cvt_f32_to_bf16(void *stream, const float *src, void *dst, int64_t n)
{
    return usot_cvt_f32_to_lp(stream, src, dst, n, 0);
}

extern "C" int usot_maxpool3x3s2_lp(void *stream, const void *x, void *y, int N, int H, int W, int C, int OH, int OW, int dtype)
{
    if (!x || !y || N <= 0 || (C & 7)) return USOT_EINVAL;
    if (OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1) return USOT_EINVAL;
    const long total = (long)N * OH * OW * (C / 8);
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    if (dtype) hipLaunchKernelGGL(maxpool3x3s2_bf16_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                  (const uint16_t *)x, (uint16_t *)y, N, H, W, C / 8, OH, OW);
    else       hipLaunchKernelGGL(maxpool3x3s2_bf16_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                  (const uint16_t *)x, (uint16_t *)y, N, H, W, C / 8, OH, OW);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_maxpool3x3s2_bf16(void *stream, const void *x, void *y, int N, int H, int W, int C, int OH, int OW)
{
    return usot_maxpool3x3s2_lp(stream, x, y, N, H, W, C, OH, OW, 0);
}
