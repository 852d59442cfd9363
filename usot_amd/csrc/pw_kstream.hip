// Channel-REDUCING 1x1 convolutions of the batched low-precision backbone: a bottleneck's conv1 + BN + ReLU in layer3
// (1024 -> 256, modules.py:40-42) and the neck's 1x1 + BN (connect.py:294-300), M = 61 504 pixels at batch 64.
//
// 157 MB of compulsory traffic for 32 GFLOP (34 us of HBM time, 13 of MFMA): the tiled implicit GEMM streams X through a
// two-stage LDS pipeline — one 32 KB k-tile in flight per CU, one HBM round trip per k-tile — and takes 47 us.  Here the
// ACCUMULATORS are stationary and K streams:
//   * a workgroup (8 waves, one per CU) owns a panel of 256 pixels; a wave keeps the whole 256-channel x 32-pixel output in
//     128 accumulator registers;
//   * X never touches LDS: a lane fetches its MFMA B fragments (16 bytes: 8 consecutive k of one pixel) straight from
//     global memory, THREE 64-deep k-chunks ahead (96 KB in flight per CU);
//   * W streams through LDS in slabs of 256 channels x 64 k (32 KB, LDS-DMA, ring of three), read by all eight waves; an A
//     fragment feeds two MFMAs (the wave's two pixel blocks);
//   * every vector-memory operation of the loop is inline asm with counted s_waitcnt vmcnt(N): both kinds of loads of chunk
//     c+2 are issued after the barrier of chunk c, so "at most the 2 x 8 youngest outstanding" means chunk c has landed;
//   * slab rows are permuted (as csrc/pw_panel.hip) so that a lane's accumulators of a block pair are 8 contiguous
//     channels: the epilogue stores 16-byte pieces, 64 contiguous bytes per pixel and instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

#ifdef USOT_KS_NT          // non-temporal policy for the activation stream (read once): scripts/probes/stride_probe.hip reads 6.9 vs 6.2 TB/s
#define USOT_KS_NTS " nt"
#else
#define USOT_KS_NTS ""
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct KStreamK {
    const uint16_t *x, *w;
    const float *bias;
    uint16_t *y;
    int M, act, npanels;
};

template <bool F16> __device__ __forceinline__ f32x4 ks_mfma(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else               return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int KS_NW = 8;            // wavefronts
constexpr int KS_BM = 256;          // pixels per panel (32 per wave)
constexpr int KS_S = 3;             // slabs / fragment sets in flight

template <int K, int N, bool F16>
__global__ __launch_bounds__(512) void pw_kstream_kernel(const KStreamK p)
{
    constexpr int NC = K / 64;                      // k-chunks
    constexpr int NB = N / 16;                      // 16-channel MFMA blocks
    constexpr int SLAB = N * 8;                     // 16-byte chunks per slab (N rows x 64 k)
    constexpr int NI = SLAB / 512;                  // DMA instructions per thread and slab
    constexpr int NV = 4 + NI;                      // vector-memory operations per wave and chunk
    static_assert(K % 64 == 0 && N % 64 == 0 && SLAB % 512 == 0, "shape");
    // LDS: [bias: N floats, padded to HEAD chunks][KS_S][SLAB].  The instruction offset of an LDS-DMA load is added to the LDS
    // address as well as to the global one, so M0 carries the destination MINUS that offset: the ring starts HEAD >= NC * 8
    // chunks in, which keeps M0 non-negative
    constexpr int HEAD = (N / 4 > NC * 8 ? N / 4 : NC * 8);
    static_assert(HEAD * 16 >= (NC - 1) * 128, "M0 = LDS destination - instruction offset must not go negative (dynamic LDS starts at 0)");
    static_assert((NC - 1) * 128 + 64 < 4096, "the chunk's k offset rides in the 12-bit instruction offset");
    extern __shared__ __attribute__((aligned(16))) u32x4 ks_lds[];
    u32x4 *ks_smem = ks_lds + HEAD;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    float *sBias = (float *)ks_lds;
    for (int i = tid; i < N; i += 512) sBias[i] = p.bias ? p.bias[i] : 0.0f;

    // slab row (block b, MFMA row rho) <-> channel (b >> 1) * 32 + (rho >> 2) * 8 + (b & 1) * 4 + (rho & 3): the lane's
    // accumulators (rows q*4 .. q*4+3) of blocks 2i, 2i+1 are channels i*32 + q*8 .. +7
    const uint16_t *wsrc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = i * 512 + tid, row = c >> 3, pc = c & 7;
        const int b = row >> 4, rho = row & 15;
        const int ch = (b >> 1) * 32 + (rho >> 2) * 8 + (b & 1) * 4 + (rho & 3);
        wsrc[i] = p.w + (long)ch * K + (pc ^ ((row >> 1) & 7)) * 8;
    }
    auto issue_w = [&](auto cc, int buf) {                       // the chunk's k offset rides in the instruction (c * 128 bytes < 4 KB)
        constexpr int c = decltype(cc)::value;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t lds = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(ks_smem + buf * SLAB + i * 512 + wave * 64)) - c * 128;
            const uint16_t *src = wsrc[i];
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(lds), "n"(c * 128) : "memory");
        }
    };
    // the lane's two A-fragment offsets in a slab (k-step 0 / 1): row l15 of a block, chunk (ks*4 + q) ^ swizzle(row)
    const int aoff0 = l15 * 8 + ((0 * 4 + q) ^ ((l15 >> 1) & 7)), aoff1 = l15 * 8 + ((1 * 4 + q) ^ ((l15 >> 1) & 7));

    for (int panel = blockIdx.x; panel < p.npanels; panel += gridDim.x) {
        const long pm0 = (long)panel * KS_BM + wave * 32;
        const uint16_t *xrow[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) xrow[j] = p.x + min(pm0 + j * 16 + l15, (long)p.M - 1) * K + q * 8;
        u32x4 xf[KS_S][2][2];                                   // [set][k-step][pixel block]
        auto issue_x = [&](auto cc, auto sc) {
            constexpr int c = decltype(cc)::value, s = decltype(sc)::value;
            u32x4 (&xs)[2][2] = xf[s];                           // (asm operands alone do not capture in a generic lambda)
            const uint16_t *x0 = xrow[0], *x1 = xrow[1];
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" USOT_KS_NTS : "=v"(xs[0][0]) : "v"(x0), "n"(c * 128) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" USOT_KS_NTS : "=v"(xs[0][1]) : "v"(x1), "n"(c * 128) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" USOT_KS_NTS : "=v"(xs[1][0]) : "v"(x0), "n"(c * 128 + 64) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" USOT_KS_NTS : "=v"(xs[1][1]) : "v"(x1), "n"(c * 128 + 64) : "memory");
        };
        f32x4 acc[NB][2];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b][0] = acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f};

        __syncthreads();                                         // every wave is done with the previous panel's slabs (and sBias is written)
        static_for<2>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (c < NC) { issue_x(cc, cc); issue_w(cc, c); }
        });
        static_for<NC>([&](auto cc) {
            constexpr int c = decltype(cc)::value, s = c % KS_S;
            // chunk c has landed when at most the operations of chunk c+1 are outstanding; the fragment registers pass through
            // the wait so that no MFMA is scheduled above it
            u32x4 (&xs)[2][2] = xf[s];
            if constexpr (c + 1 < NC)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[1][0]), "+v"(xs[1][1]) : "n"(NV) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[1][0]), "+v"(xs[1][1]) :: "memory");
            __builtin_amdgcn_s_barrier();                        // ... for every wave's slab pieces; the slab of chunk c-1 is free
            asm volatile("" ::: "memory");
            if constexpr (c + 2 < NC) {
                issue_x(std::integral_constant<int, c + 2>{}, std::integral_constant<int, (c + 2) % KS_S>{});
                issue_w(std::integral_constant<int, c + 2>{}, (c + 2) % KS_S);
            }
            const u32x4 *sl = ks_smem + s * SLAB;
            // A fragments through a pinned ring of two groups of four (left alone hipcc hoists the 32 reads of a chunk and spills)
            constexpr int NG = 2 * NB / 4;
            u32x4 ar[2][4];
            auto rd = [&](int g, u32x4 (&a)[4]) {
                const int ks = g / (NB / 4), b0 = (g % (NB / 4)) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = sl[(b0 + e) * 128 + (ks ? aoff1 : aoff0)];
            };
            rd(0, ar[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) rd(g + 1, ar[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int ks = g / (NB / 4), b0 = (g % (NB / 4)) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[b0 + e][0] = ks_mfma<F16>(ar[g & 1][e], xs[ks][0], acc[b0 + e][0]);
                    acc[b0 + e][1] = ks_mfma<F16>(ar[g & 1][e], xs[ks][1], acc[b0 + e][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });

        // ---- epilogue from registers: acc[2i + e][j][r] = channel i*32 + q*8 + e*4 + r of pixel pm0 + j*16 + l15
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long m = pm0 + j * 16 + l15;
#pragma unroll
            for (int i = 0; i < NB / 2; ++i) {
                const f32x4 b0 = *(const f32x4 *)(sBias + i * 32 + q * 8), b1 = *(const f32x4 *)(sBias + i * 32 + q * 8 + 4);
                f32x4 v0 = acc[2 * i][j] + b0, v1 = acc[2 * i + 1][j] + b1;
                if (p.act == USOT_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v0[r] = fmaxf(v0[r], 0.0f); v1[r] = fmaxf(v1[r], 0.0f); }
                }
                u32x4 o;
                o[0] = usot_pack2_lp<F16>(v0[0], v0[1]); o[1] = usot_pack2_lp<F16>(v0[2], v0[3]);
                o[2] = usot_pack2_lp<F16>(v1[0], v1[1]); o[3] = usot_pack2_lp<F16>(v1[2], v1[3]);
                if (m < p.M) *(u32x4 *)(p.y + m * N + i * 32 + q * 8) = o;
            }
        }
    }
}

template <int K, int N>
int ks_launch(void *stream, const KStreamK &p, int dtype)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    constexpr int lds = KS_S * N * 8 * 16 + ((N / 4 > (K / 64) * 8 ? N / 4 : (K / 64) * 8)) * 16;
    static bool raised_d[USOT_MAX_DEV][2] = {};
    bool (&raised)[2] = raised_d[usot_dv];
    const void *fn = dtype ? (const void *)pw_kstream_kernel<K, N, true> : (const void *)pw_kstream_kernel<K, N, false>;
    if (!raised[dtype]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return USOT_ELAUNCH;
        raised[dtype] = true;
    }
    static int cus_d[USOT_MAX_DEV] = {};
    int &cus = cus_d[usot_dv];
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    const int grid = p.npanels < cus ? p.npanels : cus;
    if (dtype) hipLaunchKernelGGL((pw_kstream_kernel<K, N, true>), dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
    else       hipLaunchKernelGGL((pw_kstream_kernel<K, N, false>), dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

}  // namespace

extern "C" int usot_pw_kstream_supported(int K, int N) { return K == 1024 && N == 256; }

/* y[M][N] = act(x[M][K] . w^T + bias): x, w ([N][K], the conv kernels' layout), y in the storage type (dtype 0 = bf16,
 * 1 = fp16), bias fp32 or NULL, act USOT_ACT_NONE | USOT_ACT_RELU; shapes: usot_pw_kstream_supported(K, N).
 * Replaces the 1024 -> 256 reductions of layer3 (modules.py:40-42) and the neck's 1x1 (connect.py:294-300) at large M. */
extern "C" int usot_pw_kstream_lp(void *stream, const void *x, const void *w, const float *bias, void *y,
                                  long M, int K, int N, int act, int dtype)
{
    if (!x || !w || !y || M <= 0 || M > 0x7fffffffL || (dtype != 0 && dtype != 1) || !usot_pw_kstream_supported(K, N)) return USOT_EINVAL;
    if (act != USOT_ACT_NONE && act != USOT_ACT_RELU) return USOT_EINVAL;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)y) & 15) return USOT_EINVAL;
    KStreamK p;
    p.x = (const uint16_t *)x; p.w = (const uint16_t *)w; p.bias = bias; p.y = (uint16_t *)y;
    p.M = (int)M; p.act = act; p.npanels = (int)((M + KS_BM - 1) / KS_BM);
    return ks_launch<1024, 256>(stream, p, dtype);
}
