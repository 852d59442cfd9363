// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, of the batched low-precision backbone (layer1's three conv2 +
// BN + ReLU, modules.py:43-46, at batch 64: M = 254 016 pixels), as a DIRECT convolution from an LDS halo tile.
//
// The tiled implicit-GEMM kernel gathers every output pixel's 9 taps from L2: 292 MB of L2 -> LDS traffic per launch for
// 65 MB of HBM bytes, at the ~9.7 TB/s that path delivers: 45 us for a layer whose bytes would take 14 and whose MFMAs 8.
// Here a workgroup (8 waves, persistent) owns a 16 x 16 spatial tile of one image:
//   * the 18 x 18 x 64 halo tile goes to LDS ONCE (LDS-DMA, double-buffered: the next tile lands under the MFMAs); a B
//     fragment of tap (kh, kw) is the same LDS image read at a shifted pixel offset — an immediate in the ds_read;
//   * the filter bank (64 x 576: 72 KB) never touches LDS: wave (h, rq) owns 32 output channels x 4 tile rows and keeps its
//     A fragments — 2 channel blocks x 18 k-steps — in 144 REGISTERS for the whole kernel; a B fragment feeds two MFMAs;
//   * LDS layout: four planes (one per lane quad q) of [pixel][3 chunks] (2 used: k-steps 0 / 1 of that quad's 8 channels,
//     1 pad): a lane group of one ds_read_b128 covers 16 consecutive pixels x {q, q+1} and 3 * pixel mod 16 is a
//     permutation, planes start on multiples of 256 B: conflict-free for every tap shift;
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
constexpr int HPLANE = 1024;            // chunks per quad plane (324 * 3 = 972 used; a multiple of 16 chunks)
constexpr int HBUF = 4 * HPLANE;        // chunks per halo buffer (64 KB)
constexpr int HNI = HBUF / 512;         // DMA instructions per thread and tile

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

    // ---- A fragments: W[channel][k], k = (kh * 3 + kw) * 64 + ci.  MFMA row rho of block i <-> channel h*32 + (rho >> 2)*8 +
    // i*4 + (rho & 3): the lane's accumulators (rows q*4 .. q*4+3 of blocks 0, 1) are channels h*32 + q*8 .. +7
    u32x4 af[2][18];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = h * 32 + (l15 >> 2) * 8 + i * 4 + (l15 & 3);
#pragma unroll
        for (int s = 0; s < 18; ++s) af[i][s] = *(const u32x4 *)(p.w + (long)ch * 576 + s * 32 + q * 8);
    }
    f32x4 bv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bv[i] = p.bias ? *(const f32x4 *)(p.bias + h * 32 + q * 8 + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};

    // halo tile of `tile` -> buffer `buf` by LDS-DMA: physical chunk c = i * 512 + tid = plane (c >> 10), pixel P and slot pc
    // of it; plane q', slot pc hold the pixel's 16-byte chunk pc * 4 + q' (channels (pc * 4 + q') * 8 .. +7)
    auto issue_halo = [&](int tile, int buf) {
        const int n = tile / (p.tiles_x * p.tiles_y), r2 = tile - n * p.tiles_x * p.tiles_y;
        const int ty = r2 / p.tiles_x, tx = r2 - ty * p.tiles_x;
        const int y0 = ty * HT - 1, x0 = tx * HT - 1;
#pragma unroll
        for (int i = 0; i < HNI; ++i) {
            const int c = i * 512 + tid;
            const int plane = c >> 10, idx = c & (HPLANE - 1);
            const int P = idx / 3, pc = idx - P * 3;
            const int hy = P / HH, hx = P - hy * HH;
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = P < HPIX && pc < 2 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint16_t *src = ok ? p.x + (((long)n * p.H + iy) * p.W + ix) * 64 + (pc * 4 + plane) * 8 : (const uint16_t *)h_zero16;
            const uint32_t lds = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(h_smem + buf * HBUF + i * 512 + wave * 64));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
        }
    };

    int tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    issue_halo(tile, 0);
    int buf = 0;
    // the lane's base chunk in a buffer: plane q, pixel (rq * 4) * 18 + l15 (tile row rq*4, column l15, before the tap shift)
    const int lane_chunk = q * HPLANE + ((rq * 4) * HH + l15) * 3;
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
        if (next < p.ntiles) issue_halo(next, buf ^ 1);         // in flight under the MFMAs below
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
            for (int j = 0; j < 4; ++j) b[j] = hb[((j + kh) * HH + kw) * 3 + ks];
        };
        read_b(0, bf[0]);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            if (s + 1 < 18) read_b(s + 1, bf[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = h_mfma<F16>(af[i][s], bf[s & 1][j], acc[i][j]);
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
            if (oy < p.H && ox < p.W) {
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
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    constexpr int lds = 2 * HBUF * 16;
    static bool raised[2] = {false, false};
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
