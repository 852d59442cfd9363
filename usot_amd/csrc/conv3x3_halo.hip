// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, of the batched low-precision backbone (layer1's three conv2 +
// BN + ReLU, modules.py:43-46, at batch 64: M = 254 016 pixels), as a DIRECT convolution from an LDS halo tile.
//
// The tiled implicit-GEMM kernel gathers every output pixel's 9 taps from L2: 292 MB of L2 -> LDS traffic per launch for
// 65 MB of HBM bytes, at the ~9.7 TB/s that path delivers: 45 us for a layer whose bytes would take 14 and whose MFMAs 8.
// Here a workgroup (8 waves, persistent) owns a 16 x 16 spatial tile of one image:
//   * the 18 x 18 x 64 halo tile goes to LDS ONCE (LDS-DMA, double-buffered: the next tile lands under the MFMAs); a B
//     fragment of tap (kh, kw) is the same LDS image read at a shifted pixel offset — an immediate in the ds_read;
//   * the filter bank (64 x 576: 72 KB) is read ONCE per workgroup: wave (h, rq) owns 32 output channels x 4 tile rows and
//     keeps its A fragments — 2 channel blocks x 18 k-steps — in 144 REGISTERS for the whole kernel; a B fragment feeds two
//     MFMAs.  The four waves of a channel half hold the same fragments: the bank goes through LDS in fragment order (DMA
//     with per-lane sources, beside the first halo, in the space of the second halo buffer) and every wave reads its 36
//     fragments linearly — fetching them from global memory per wave cost 2 us of the launch;
//   * LDS layout: [pixel][10 chunks of 16 B] — the pixel's 128 contiguous bytes + 2 pad chunks, so a DMA instruction copies
//     whole 128-byte lines (6.4 pixels).  ds_read_b128 is served in lane groups {l15 0-3, 12-15 of quad q ; l15 4-11 of quad
//     q+1}: 10 * l15 mod 16 takes the eight even slots on either part and the quads are one chunk apart: conflict-free for
//     every tap shift (the four-plane [pixel][3] layout of the first version was too, but its DMA gathered 16-byte pieces:
//     2.5 us per tile of DMA alone against 3.5 us of HBM time);
//   * rows of the filter bank are permuted (as csrc/pw_panel.hip) so that a lane's 8 accumulator registers per pixel are 8
//     contiguous channels: the epilogue stores 16-byte pieces, 64 contiguous bytes per pixel and instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct HaloK {
    const uint16_t *x, *w;
    const float *bias;
    uint16_t *y;
    int N, H, W, act, tiles_x, tiles_y, ntiles;
};

__device__ __attribute__((aligned(16))) uint32_t h_zero16[4] = {0u, 0u, 0u, 0u};     // source of out-of-image pixels and pad chunks

constexpr int HT = 16;                  // tile edge (output pixels)
constexpr int HH = HT + 2;              // halo edge
constexpr int HPIX = HH * HH;           // 324 halo pixels
constexpr int HPS = 10;                 // chunks per halo pixel in LDS (8 data + 2 pad)
constexpr int HNI = (HPIX * HPS + 511) / 512;   // DMA instructions per thread and tile (7)
constexpr int HBUF = HNI * 512;         // chunks per halo buffer (56 KB)
constexpr int HWCH = 64 * 576 / 8;      // chunks of the filter bank (4608: 72 KB), staged once at [HBUF, HBUF + HWCH)
constexpr int HLDS = (HBUF + HWCH > 2 * HBUF ? HBUF + HWCH : 2 * HBUF) * 16;

template <bool F16> __device__ __forceinline__ f32x4 h_mfma(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else               return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool F16>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const HaloK p)
{
    extern __shared__ __attribute__((aligned(16))) u32x4 h_smem[];      // [2][HBUF]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    const int h = wave & 1, rq = wave >> 1;             // channel half, row quarter of the tile

    auto dma16 = [&](const void *src, int chunk0) {             // 64 lanes x 16 B -> LDS chunks chunk0 + lane
        const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(h_smem + chunk0));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
    };

    // halo tile of `tile` -> buffer `buf` by LDS-DMA: chunk c = i * 512 + tid = pixel P = c / 10, slot pc of it (8, 9: pad)
    auto issue_halo = [&](int tile, int buf) {
        const int n = tile / (p.tiles_x * p.tiles_y), r2 = tile - n * p.tiles_x * p.tiles_y;
        const int ty = r2 / p.tiles_x, tx = r2 - ty * p.tiles_x;
        const int y0 = ty * HT - 1, x0 = tx * HT - 1;
#pragma unroll
        for (int i = 0; i < HNI; ++i) {
            const int c = i * 512 + tid;
            const int P = c / HPS, pc = c - P * HPS;
            const int hy = P / HH, hx = P - hy * HH;
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = P < HPIX && pc < 8 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint16_t *src = ok ? p.x + (((long)n * p.H + iy) * p.W + ix) * 64 + pc * 8 : (const uint16_t *)h_zero16;
            dma16(src, buf * HBUF + i * 512 + wave * 64);
        }
    };

    int tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    issue_halo(tile, 0);
    // ---- A fragments: W[channel][k], k = (kh * 3 + kw) * 64 + ci.  MFMA row rho of block i <-> channel h*32 + (rho >> 2)*8 +
    // i*4 + (rho & 3): the lane's accumulators (rows q*4 .. q*4+3 of blocks 0, 1) are channels h*32 + q*8 .. +7.  Staged in
    // fragment order: chunk ((h*2 + i) * 18 + s) * 64 + lane
#pragma unroll
    for (int i = 0; i < HWCH / 512; ++i) {
        const int c = i * 512 + tid;
        const int ln = c & 63, fs = c >> 6, s = fs % 18, hi = fs / 18;
        const int ch = (hi >> 1) * 32 + ((ln & 15) >> 2) * 8 + (hi & 1) * 4 + (ln & 3);
        dma16(p.w + (long)ch * 576 + s * 32 + (ln >> 4) * 8, HBUF + i * 512 + wave * 64);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u32x4 af[2][18];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 18; ++s) af[i][s] = h_smem[HBUF + ((h * 2 + i) * 18 + s) * 64 + lane];
    f32x4 bv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bv[i] = p.bias ? *(const f32x4 *)(p.bias + h * 32 + q * 8 + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    // (the barrier at the top of the first tile orders these reads before the second halo lands on the staging space)
    int buf = 0;
    // the lane's base chunk in a buffer: pixel (rq * 4) * 18 + l15 (tile row rq*4, column l15, before the tap shift), quad q
    const int lane_chunk = ((rq * 4) * HH + l15) * HPS + q;
    int stores_prev = -1;                                       // stores this wave issued after the newest DMAs (-1: none yet)
    for (; tile < p.ntiles; tile += gridDim.x) {
        // this tile's halo must have landed; the only younger vector-memory operations are the previous tile's stores (one
        // per tile row of this wave that exists: a wave-uniform count), which need not be waited for — vmcnt retires in order
        switch (stores_prev) {
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        __syncthreads();                                        // ... for every wave; the other buffer's readers are done
        const int next = tile + gridDim.x;
#ifndef USOT_HLABL_NODMA
        if (next < p.ntiles) issue_halo(next, buf ^ 1);         // in flight under the MFMAs below
#endif
        const u32x4 *hb = h_smem + buf * HBUF + lane_chunk;

        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // B fragments through a ring of two k-steps, pinned (left alone hipcc hoists dozens of the 72 reads and spills):
        // the reads of k-step s+1 are issued before the MFMAs of k-step s
        u32x4 bf[2][4];
        auto read_b = [&](int s, u32x4 (&b)[4]) {
            const int t = s >> 1, ks = s & 1, kh = t / 3, kw = t - kh * 3;
#pragma unroll
#ifdef USOT_HLABL_NOLDS
            for (int j = 0; j < 4; ++j) b[j] = af[1][17 - s];
#else
            for (int j = 0; j < 4; ++j) b[j] = hb[((j + kh) * HH + kw) * HPS + ks * 4];
#endif
        };
        read_b(0, bf[0]);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            if (s + 1 < 18) read_b(s + 1, bf[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#ifdef USOT_HLABL_NOMMA
                    if (s < 2) acc[i][j] = h_mfma<F16>(af[i][s], bf[s & 1][j], acc[i][j]);
                    else acc[i][j][0] += __builtin_bit_cast(float, af[i][s][0] ^ bf[s & 1][j][0]);
#else
                    acc[i][j] = h_mfma<F16>(af[i][s], bf[s & 1][j], acc[i][j]);
#endif
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: acc[i][j][r] = channel h*32 + q*8 + i*4 + r of tile pixel (rq*4 + j, l15)
        const int n = tile / (p.tiles_x * p.tiles_y), r2 = tile - n * p.tiles_x * p.tiles_y;
        const int ty = r2 / p.tiles_x, tx = r2 - ty * p.tiles_x;
        const int ox = tx * HT + l15;
        stores_prev = min(4, max(0, p.H - (ty * HT + rq * 4)));  // rows of this wave inside the image (column 0 of a tile always is)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = ty * HT + rq * 4 + j;
            float v[8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[i * 4 + r] = acc[i][j][r] + bv[i][r];
                    if (p.act == USOT_ACT_RELU) v[i * 4 + r] = fmaxf(v[i * 4 + r], 0.0f);
                }
#ifdef USOT_HLABL_NOSTORE
            if (oy < p.H && ox < p.W && v[0] == 1234.5f) {
#else
            if (oy < p.H && ox < p.W) {
#endif
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = usot_pack2_lp<F16>(v[2 * e], v[2 * e + 1]);
                *(u32x4 *)(p.y + (((long)n * p.H + oy) * p.W + ox) * 64 + h * 32 + q * 8) = o;
            }
        }
        buf ^= 1;
    }
}

}  // namespace

extern "C" int usot_conv3x3_halo_supported(int Cin, int Cout) { return Cin == 64 && Cout == 64; }

/* y = act(conv3x3(x, w) + bias), stride 1, pad 1, dilation 1, NHWC dense, x / w / y in the storage type (dtype 0 = bf16,
 * 1 = fp16), w [Cout][9 * Cin] with k = (kh * 3 + kw) * Cin + ci (the conv kernels' layout), bias fp32 or NULL,
 * act USOT_ACT_NONE | USOT_ACT_RELU.  Shapes: usot_conv3x3_halo_supported(Cin, Cout). */
extern "C" int usot_conv3x3_halo_lp(void *stream, const void *x, const void *w, const float *bias, void *y,
                                    int N, int H, int W, int Cin, int Cout, int act, int dtype)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || (dtype != 0 && dtype != 1) || !usot_conv3x3_halo_supported(Cin, Cout)) return USOT_EINVAL;
    if (act != USOT_ACT_NONE && act != USOT_ACT_RELU) return USOT_EINVAL;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)y) & 15) return USOT_EINVAL;
    HaloK p;
    p.x = (const uint16_t *)x; p.w = (const uint16_t *)w; p.bias = bias; p.y = (uint16_t *)y;
    p.N = N; p.H = H; p.W = W; p.act = act;
    p.tiles_x = usot_cdiv(W, HT); p.tiles_y = usot_cdiv(H, HT);
    const long nt = (long)p.tiles_x * p.tiles_y * N;
    if (nt > 0x7fffffffL) return USOT_EINVAL;
    p.ntiles = (int)nt;
    static int cus_d[USOT_MAX_DEV] = {};
    int &cus = cus_d[usot_dv];
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    constexpr int lds = HLDS;
    static bool raised_d[USOT_MAX_DEV][2] = {};
    bool (&raised)[2] = raised_d[usot_dv];
    const void *fn = dtype ? (const void *)conv3x3_halo_kernel<true> : (const void *)conv3x3_halo_kernel<false>;
    if (!raised[dtype]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return USOT_ELAUNCH;
        raised[dtype] = true;
    }
    const int grid = p.ntiles < cus ? p.ntiles : cus;
    if (dtype) hipLaunchKernelGGL(conv3x3_halo_kernel<true>, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
    else       hipLaunchKernelGGL(conv3x3_halo_kernel<false>, dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}
