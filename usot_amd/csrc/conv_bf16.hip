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
__global__ __launch_bounds__(256, 2) void conv_igemm_bf16(const ConvB p)
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

    // Epilogue.  The accumulator layout gives a lane four channels of one pixel: stored
    // directly that is 8-byte pieces scattered over 16 pixel rows per instruction, and the
    // residual is read the same way (measured 1.1-1.4 TB/s on the 1x1 expansion convs, which
    // are pure HBM traffic).  Instead the fp32 tile is transposed through LDS (free after the
    // k-loop) so that every lane handles 8 consecutive channels: 16-byte residual loads and
    // 16-byte stores, 16 lanes per 256-byte row segment.
    constexpr int OP = BN + 4;                       // fp32 row pitch: 4*odd dwords -> conflict-free
    if (!p.out_f32 && (p.Cout & 7) == 0) {
        float *sO = (float *)smem4;
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i)
                *(f32x4 *)(sO + ((wm * TM + j) * 16 + l15) * OP + (wn * TN + i) * 16 + quad * 4) = acc[i][j];
        __syncthreads();
        constexpr int CPRW = BN / 8;                 // 16-byte output chunks per row
        constexpr int RPASS = 256 / CPRW;            // rows per pass
        constexpr int NP = (BM + RPASS - 1) / RPASS;
        const int oc = tid % CPRW, orow = tid / CPRW;
        const int co = bn0 + oc * 8;
        const bool cok = co < p.Cout;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (p.bias && cok) { b0 = *(const f32x4 *)(p.bias + co); b1 = *(const f32x4 *)(p.bias + co + 4); }
        constexpr int CH = NP < 4 ? NP : 4;          // residual loads in flight per lane
#pragma unroll 1
        for (int q0 = 0; q0 < NP; q0 += CH) {
            u32x4 rr[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int r = orow + (q0 + q) * RPASS, m = bm0 + r;
                rr[q] = u32x4{0u, 0u, 0u, 0u};
                if (p.res && cok && r < BM && m < p.M) rr[q] = *(const u32x4 *)(p.res + (long)m * p.Cout + co);
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int r = orow + (q0 + q) * RPASS, m = bm0 + r;
                if (!cok || r >= BM || m >= p.M) continue;
                f32x4 v0 = *(const f32x4 *)(sO + r * OP + oc * 8) + b0;
                f32x4 v1 = *(const f32x4 *)(sO + r * OP + oc * 8 + 4) + b1;
                if (p.res) {
                    v0[0] += unpack_lp<F16>(rr[q][0] & 0xffffu); v0[1] += unpack_lp<F16>(rr[q][0] >> 16);
                    v0[2] += unpack_lp<F16>(rr[q][1] & 0xffffu); v0[3] += unpack_lp<F16>(rr[q][1] >> 16);
                    v1[0] += unpack_lp<F16>(rr[q][2] & 0xffffu); v1[1] += unpack_lp<F16>(rr[q][2] >> 16);
                    v1[2] += unpack_lp<F16>(rr[q][3] & 0xffffu); v1[3] += unpack_lp<F16>(rr[q][3] >> 16);
                }
                if (p.act == USOT_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                }
                u32x4 o;
                o[0] = pack_lp<F16>(v0[0]) | (pack_lp<F16>(v0[1]) << 16);
                o[1] = pack_lp<F16>(v0[2]) | (pack_lp<F16>(v0[3]) << 16);
                o[2] = pack_lp<F16>(v1[0]) | (pack_lp<F16>(v1[1]) << 16);
                o[3] = pack_lp<F16>(v1[2]) | (pack_lp<F16>(v1[3]) << 16);
                *(u32x4 *)(p.y + (long)m * p.Cout + co) = o;
            }
        }
        return;
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

// ---------------------------------------------------------------------------------------
// Low-precision stem: 7x7/s2 conv (3 -> 64) + folded BN + ReLU + 3x3/s2/p1 max-pool in one
// kernel, on v_mfma_f32_16x16x32_{bf16,f16}.  modules.py:70-74,138-141.
//
// K = 147 is too ragged for the generic implicit GEMM, so the k axis is laid out as
// 24 rows x 8: row r = ci*7 + kh (21 real rows, 3 zero rows), 8 = kw 0..6 + one zero tap.
// A lane's B fragment for k-step s is then the 8 consecutive input pixels
// x[ci][2*sy+kh][2*sx .. 2*sx+7] of row r = 4*s + quad: four 4-byte LDS reads from the
// staged input patch (the eighth pixel meets a zero weight).  The crop is staged as x - mu[ci]
// (mu: constant per input channel, its product with the UNROUNDED filters is folded into the bias
// by the host): stem filters are close to zero-sum while raw 0..255 pixels carry a large mean, so
// rounding the filters to 8-11 bits would otherwise leak mean * sum(dw) into every output.  The 64x192 filter bank lives in
// registers as pre-swizzled A fragments (host layout [cblk][kstep][lane][8]).
// A workgroup owns a 4x8 tile of POOLED pixels = a 9x17 tile of stem pixels (one halo row /
// column, recomputed: x1.2) = a 23x39 patch of the crop.  Stem outputs go to LDS in the
// storage type, the pool reads them back 8 channels per lane and writes NHWC 16-byte pieces:
// the 125x125x64 stem map never reaches HBM (256 MB fp32 + 128 MB at batch 64 before).
constexpr int SP_P = 4, SP_Q = 8;                   // pooled tile
constexpr int SP_R = 2 * SP_P + 1, SP_C = 2 * SP_Q + 1;   // 9 x 17 stem pixels
constexpr int SP_NPIX = SP_R * SP_C;                // 153
constexpr int SP_NBLK = (SP_NPIX + 15) / 16;        // 10 MFMA pixel blocks
constexpr int SP_IR = 2 * SP_R + 5, SP_IC = 2 * SP_C + 5;   // 23 x 39 input patch
constexpr int SP_ICP = 40;                          // padded patch row (elements)
constexpr int SP_OPB = 144;                         // stem-tile LDS row pitch in bytes (64 ch * 2 + 16)

template <bool F16>
__global__ __launch_bounds__(256) void stem_pool_lp_kernel(
    const float *__restrict__ x, const u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
    uint16_t *__restrict__ y, int H, int W, int OH, int OW, int PH, int PW, float mu0, float mu1, float mu2)
{
    // bf16 keeps 8 significant bits: rounding the CROP to bf16 doubles the end-to-end error of the
    // whole backbone (measured 4e-2 -> 8e-2 of the feature scale), so the bf16 variant stages the
    // crop as hi + lo (two bf16, ~16 bits) and issues two MFMAs per fragment; fp16 (11 bits) does not.
    constexpr bool SPLIT = !F16;
    constexpr int PLANE = 3 * SP_IR * SP_ICP;
    __shared__ __attribute__((aligned(16))) uint16_t patch[(SPLIT ? 2 : 1) * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned char stile[SP_NBLK * 16 * SP_OPB];
    const int n = blockIdx.z;
    const int py0 = blockIdx.y * SP_P, px0 = blockIdx.x * SP_Q;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;
    const int iy0 = 2 * sy0, ix0 = 2 * sx0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, quad = lane >> 4;

    // filter fragments: 4 channel blocks x 6 k-steps, 16 bytes each, coalesced
    u32x4 wf[4][6];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) wf[cb][ks] = wfrag[(cb * 6 + ks) * 64 + lane];

    // input patch -> LDS in the storage type (out-of-image pixels and the pad column are zero;
    // they only ever feed stem pixels the pool masks out, or zero weights)
    const float *xn = x + (long)n * 3 * H * W;
    for (int i = tid; i < 3 * SP_IR * SP_ICP; i += 256) {
        const int ci = i / (SP_IR * SP_ICP), r = i - ci * SP_IR * SP_ICP;
        const int py = r / SP_ICP, px = r - py * SP_ICP;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if (px < SP_IC && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = xn[((long)ci * H + iy) * W + ix] - (ci == 0 ? mu0 : (ci == 1 ? mu1 : mu2));
        const uint32_t hi = pack_lp<F16>(v);
        patch[i] = (uint16_t)hi;
        if constexpr (SPLIT) patch[PLANE + i] = (uint16_t)pack_lp<F16>(v - unpack_lp<F16>(hi));
    }
    __syncthreads();

    f32x4 bv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bv[cb] = *(const f32x4 *)(bias + cb * 16 + quad * 4);

    for (int blk = wave; blk < SP_NBLK; blk += 4) {
        int pi = blk * 16 + l15;
        if (pi > SP_NPIX - 1) pi = SP_NPIX - 1;
        const int sy = pi / SP_C, sx = pi - sy * SP_C;
        f32x4 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            int r = 4 * ks + quad;
            if (r > 20) r = 20;                          // zero-weight rows: any valid address
            const int ci = r / 7, kh = r - ci * 7;
            const uint32_t *src = (const uint32_t *)(patch + (ci * SP_IR + 2 * sy + kh) * SP_ICP + 2 * sx);
            u32x4 xf;
            xf[0] = src[0]; xf[1] = src[1]; xf[2] = src[2]; xf[3] = src[3];
            if constexpr (F16) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[cb][ks]),
                                                                     __builtin_bit_cast(f16x8, xf), acc[cb], 0, 0, 0);
            } else {
                const uint32_t *srl = src + PLANE / 2;
                u32x4 xl;
                xl[0] = srl[0]; xl[1] = srl[1]; xl[2] = srl[2]; xl[3] = srl[3];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[cb][ks]),
                                                                      __builtin_bit_cast(bf16x8, xf), acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[cb][ks]),
                                                                      __builtin_bit_cast(bf16x8, xl), acc[cb], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x4 v = acc[cb] + bv[cb];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            u32x2 o;
            o[0] = pack_lp<F16>(v[0]) | (pack_lp<F16>(v[1]) << 16);
            o[1] = pack_lp<F16>(v[2]) | (pack_lp<F16>(v[3]) << 16);
            *(u32x2 *)(stile + (blk * 16 + l15) * SP_OPB + (cb * 16 + quad * 4) * 2) = o;
        }
    }
    __syncthreads();

    // 3x3 / stride 2 / pad 1 max-pool: one pooled pixel x 8 channels per lane
    const int pp = tid >> 3, c8 = tid & 7;
    const int ppy = pp / SP_Q, ppx = pp - ppy * SP_Q;
    const int py = py0 + ppy, px = px0 + ppx;
    if (py >= PH || px >= PW) return;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int ly = 2 * ppy + dy, gy = sy0 + ly;
        if ((unsigned)gy >= (unsigned)OH) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int lx = 2 * ppx + dx, gx = sx0 + lx;
            if ((unsigned)gx >= (unsigned)OW) continue;
            const u32x4 v = *(const u32x4 *)(stile + (ly * SP_C + lx) * SP_OPB + c8 * 16);
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
    *(u32x4 *)(y + ((((long)n * PH + py) * PW + px) * 64 + c8 * 8)) = o;
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
        (d->res && (uintptr_t)d->res % 16) || (d->bias && (uintptr_t)d->bias % 16)) return USOT_EINVAL;
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
    size_t lds = (size_t)2 * (tc.bm + tc.bn) * LDC * 16;
    const size_t lds_out = (size_t)tc.bm * (tc.bn + 4) * 4;      // fp32 staging tile of the epilogue
    if (lds_out > lds) lds = lds_out;
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

/* Fused low-precision stem + max-pool.  x NCHW fp32 [N][3][H][W]; wfrag = the 64x147 filter bank
 * (BN folded) in the storage type, pre-swizzled as MFMA A fragments [4][6][64][8] (see
 * usot_amd/engine.py: pack_stem_lp); bias fp32[64] INCLUDING sum_k w[co][k] * mu[ci(k)] (the kernel
 * convolves x - mu); y NHWC [N][PH][PW][64] in the storage type. */
extern "C" int usot_stem_pool_lp(void *stream, const float *x, const void *wfrag, const float *bias, void *y,
                                 int N, int H, int W, int OH, int OW, int PH, int PW, int dtype,
                                 float mu0, float mu1, float mu2)
{
    if (!x || !wfrag || !bias || !y || N <= 0 || H < 7 || W < 7 || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    if (OH != (H - 7) / 2 + 1 || OW != (W - 7) / 2 + 1) return USOT_EINVAL;
    if (PH != (OH + 2 - 3) / 2 + 1 || PW != (OW + 2 - 3) / 2 + 1) return USOT_EINVAL;
    if (((uintptr_t)wfrag % 16) || ((uintptr_t)y % 16) || ((uintptr_t)bias % 16) || N > 65535) return USOT_EINVAL;
    dim3 grid(usot_cdiv(PW, SP_Q), usot_cdiv(PH, SP_P), N);
    if (dtype) hipLaunchKernelGGL(stem_pool_lp_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (const u32x4 *)wfrag, bias, (uint16_t *)y, H, W, OH, OW, PH, PW, mu0, mu1, mu2);
    else       hipLaunchKernelGGL(stem_pool_lp_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (const u32x4 *)wfrag, bias, (uint16_t *)y, H, W, OH, OW, PH, PW, mu0, mu1, mu2);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}
