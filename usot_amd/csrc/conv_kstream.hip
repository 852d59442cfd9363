// 3x3 convolutions of the batched low-precision backbone with K = 9 * Cin >= 2304 (layer3's shortcut conv 512 -> 1024, its
// six conv2 256 -> 256 (dilation 1 / 2), layer2's shortcut conv 256 -> 512 / stride 2; modules.py:43-46,115-126) with the
// ACCUMULATORS stationary and K streaming — csrc/pw_kstream.hip generalised to an implicit GEMM:
//   * a workgroup (8 waves) owns 256 output pixels x 256 output channels; a wave keeps its 32 pixels x 256 channels in 128
//     accumulator registers for all of K;
//   * X never touches LDS: a lane IS a pixel — it fetches the MFMA B fragments (16 bytes: 8 consecutive input channels of its
//     pixel's tap) straight from global memory, three 64-deep k-chunks ahead; a padding tap re-reads the lane's centre
//     pixel and is zeroed once it has landed;
//   * W streams through LDS in slabs of 256 channels x 64 k (LDS-DMA, ring of three), an A fragment feeds two MFMAs;
//   * every vector-memory operation of the loop is inline asm under counted s_waitcnt vmcnt(8): the tiled kernel's
//     [DMA, fragment reads, MFMAs, vmcnt(0) + barrier] per k-tile leaves the matrix pipe 67 % busy —
//     here nothing is drained at a barrier, half the bytes of a k-step bypass the LDS, and the X queue is three chunks deep.
// k order: tap-major (k = (kh*3 + kw) * Cin + ci, the conv kernels' filter layout); the static loop body is one kernel ROW
// (3 taps x Cin/64 chunks, a multiple of the ring length), the rows are a run-time loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct CKStreamK {
    const uint16_t *x, *w;
    const float *bias;
    uint16_t *y;
    int N, H, W, OH, OW, Cout, stride, pad, dil, act, M, npanels, NT;
};

template <bool F16> __device__ __forceinline__ f32x4 ck_mfma(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else               return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ int ck_xcd_remap(int b, int total)       // consecutive tiles on ONE XCD (workgroup ids go round the 8 XCDs)
{
    const int q = total >> 3, r = total & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

constexpr int CK_BM = 256, CK_BN = 256, CK_S = 3;

template <int CIN, bool F16>
__global__ __launch_bounds__(512) void conv_kstream_kernel(const CKStreamK p)
{
    constexpr int CC = CIN / 64;                    // k-chunks per tap
    constexpr int RC = 3 * CC;                      // chunks per kernel row (the static body)
    constexpr int K = 9 * CIN;
    constexpr int NB = CK_BN / 16;
    constexpr int SLAB = CK_BN * 8;                 // 16-byte chunks per slab
    constexpr int NI = SLAB / 512;                  // DMA instructions per thread and slab (4)
    constexpr int NV = 4 + NI;
    constexpr int HEAD = (CK_BN / 4 > CC * 8 ? CK_BN / 4 : CC * 8);     // bias, and room for M0 = destination - instruction offset
    static_assert(CIN % 64 == 0 && RC % CK_S == 0 && CC >= 2 && CC * 128 + 64 + 16 <= 4096, "shape (instruction offsets are 13-bit signed)");
    static_assert(HEAD * 16 >= (CC - 1) * 128, "M0 = LDS destination - instruction offset must not go negative (dynamic LDS starts at 0)");
    extern __shared__ __attribute__((aligned(16))) u32x4 ck_lds[];
    u32x4 *slabs = ck_lds + HEAD;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    const int bb = ck_xcd_remap(blockIdx.x, p.npanels * p.NT);
    const int nt = bb % p.NT, panel = bb / p.NT;    // the NT channel tiles of a pixel panel run side by side on one XCD
    float *sBias = (float *)ck_lds;
    for (int i = tid; i < CK_BN; i += 512) sBias[i] = p.bias ? p.bias[nt * CK_BN + i] : 0.0f;

    // slab row (block b, MFMA row rho) <-> channel (b >> 1) * 32 + (rho >> 2) * 8 + (b & 1) * 4 + (rho & 3) of the tile
    uint32_t wrow[NI];                                          // byte offsets from the tap's (wave-uniform) base: saddr + voffset loads
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = i * 512 + tid, row = c >> 3, pc = c & 7;
        const int b = row >> 4, rho = row & 15;
        const int ch = (b >> 1) * 32 + (rho >> 2) * 8 + (b & 1) * 4 + (rho & 3);
        wrow[i] = (uint32_t)(ch * K + (pc ^ ((row >> 1) & 7)) * 8) * 2u;
    }
    const uint16_t *wtile = p.w + (long)nt * CK_BN * K;
    // the lane's two pixels (one per 16-pixel block of the wave).  A tap's address is a wave-uniform base (x + the tap's offset
    // from the CENTRE tap) + a per-lane byte offset that never changes: the lane's centre pixel.  A lane whose tap falls outside
    // the image points back at its own centre pixel (always inside: launcher) and zeroes the fragment after it has landed.
    const long pm0 = (long)panel * CK_BM + wave * 32;
    uint32_t xcen[2];                                           // byte offset of the centre-tap pixel's first channel of this quad
    uint32_t vmask = 0;                                         // bit (kh*3 + kw) * 2 + j: the tap is inside the image
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = (int)min(pm0 + j * 16 + l15, (long)p.M - 1);
        const int n = m / (p.OH * p.OW), r = m - n * p.OH * p.OW;
        const int oh = r / p.OW, ow = r - oh * p.OW;
        const int ihc = oh * p.stride - p.pad + p.dil, iwc = ow * p.stride - p.pad + p.dil;
        xcen[j] = (uint32_t)(((n * p.H + ihc) * p.W + iwc) * CIN + q * 8) * 2u;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ih = ihc + (t / 3 - 1) * p.dil, iw = iwc + (t % 3 - 1) * p.dil;
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) vmask |= 1u << (t * 2 + j);
        }
    }
    uint32_t xoff[2];                                           // the issue side's current per-lane offsets (two chunks ahead of the MFMAs)
    const uint16_t *xbase = p.x, *wtap = wtile;                 // ... and its wave-uniform bases
    auto set_tap = [&](int kh, int kw) {
        const int rel = ((kh - 1) * p.dil * p.W + (kw - 1) * p.dil) * CIN;     // elements from the centre tap
        xbase = p.x + rel;
#pragma unroll
        for (int j = 0; j < 2; ++j) {                           // parked: the centre pixel, or the base itself where that lies before it
            const long park = (long)xcen[j] - 2L * rel;
            xoff[j] = ((vmask >> ((kh * 3 + kw) * 2 + j)) & 1u) ? xcen[j] : (uint32_t)(park > 0 ? park : 0);
        }
        wtap = wtile + (kh * 3 + kw) * CIN;
    };
    u32x4 xf[CK_S][2][2];                                       // [set][k-step][pixel block]
    auto issue = [&](auto ccc, auto sc, int buf) {               // chunk cc of the current tap -> fragment set s, slab buf
        constexpr int cc = decltype(ccc)::value, s = decltype(sc)::value;
        u32x4 (&xs)[2][2] = xf[s];
        const uint32_t x0 = xoff[0], x1 = xoff[1];
        const uint16_t *xb = xbase;
        asm volatile("global_load_dwordx4 %0, %1, %3 offset:%2" : "=v"(xs[0][0]) : "v"(x0), "n"(cc * 128), "s"(xb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %3 offset:%2" : "=v"(xs[0][1]) : "v"(x1), "n"(cc * 128), "s"(xb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %3 offset:%2" : "=v"(xs[1][0]) : "v"(x0), "n"(cc * 128 + 64), "s"(xb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %3 offset:%2" : "=v"(xs[1][1]) : "v"(x1), "n"(cc * 128 + 64), "s"(xb) : "memory");
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            // (the instruction offset of an LDS-DMA load is added to the LDS address too: M0 = destination - offset)
            const uint32_t lds = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(slabs + buf * SLAB + i * 512 + wave * 64)) - cc * 128;
            const uint32_t voff = wrow[i];
            const uint16_t *sbase = wtap;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4 offset:%3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(lds), "n"(cc * 128), "s"(sbase) : "memory");
        }
    };
    const int aoff0 = l15 * 8 + ((0 * 4 + q) ^ ((l15 >> 1) & 7)), aoff1 = l15 * 8 + ((1 * 4 + q) ^ ((l15 >> 1) & 7));
    f32x4 acc[NB][2];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b][0] = acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();
    set_tap(0, 0);
    issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0);
    issue(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, 1);
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
        static_for<RC>([&](auto cidx) {
            constexpr int c = decltype(cidx)::value, s = c % CK_S;
            u32x4 (&xs)[2][2] = xf[s];
            // chunk (kh, c) has landed when at most the NV operations of the next chunk are outstanding.  The body is the same
            // for every chunk: past the end of K the issue side re-reads the last kernel row's first two chunks (2 of 9 * CC
            // chunks of extra traffic) instead of branching — a conditional definition of asm-loaded registers would leave
            // the compiler free to copy a register whose load is still in flight
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[1][0]), "+v"(xs[1][1]) : "n"(NV) : "memory");
            {                                                    // padding taps: the fragment that came from the centre pixel becomes zero
                constexpr int kw = c / CC;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool ok = (vmask >> ((kh * 3 + kw) * 2 + j)) & 1u;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int e = 0; e < 4; ++e) xs[ks][j][e] = ok ? xs[ks][j][e] : 0u;
                }
            }
            __builtin_amdgcn_s_barrier();                        // ... for every wave's slab pieces; the slab of the previous chunk is free
            asm volatile("" ::: "memory");
            {                                                    // issue chunk c + 2 (possibly the next kernel row's)
                constexpr int c2 = (c + 2) % RC, kw2 = c2 / CC, cc2 = c2 % CC;
                constexpr bool wraps = c + 2 >= RC;
                if constexpr (cc2 == 0) set_tap(wraps ? min(kh + 1, 2) : kh, kw2);
                issue(std::integral_constant<int, cc2>{}, std::integral_constant<int, (c + 2) % CK_S>{}, (c + 2) % CK_S);
            }
            const u32x4 *sl = slabs + s * SLAB;
            constexpr int GB = 2;                                // A fragments per ring slot (two slots: 16 registers)
            constexpr int NG = 2 * NB / GB;
            u32x4 ar[2][GB];
            auto rd = [&](int g, u32x4 (&a)[GB]) {
                const int ks = g / (NB / GB), b0 = (g % (NB / GB)) * GB;
#pragma unroll
                for (int e = 0; e < GB; ++e) a[e] = sl[(b0 + e) * 128 + (ks ? aoff1 : aoff0)];
            };
            rd(0, ar[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) rd(g + 1, ar[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const int ks = g / (NB / GB), b0 = (g % (NB / GB)) * GB;
#pragma unroll
                for (int e = 0; e < GB; ++e) {
                    acc[b0 + e][0] = ck_mfma<F16>(ar[g & 1][e], xs[ks][0], acc[b0 + e][0]);
                    acc[b0 + e][1] = ck_mfma<F16>(ar[g & 1][e], xs[ks][1], acc[b0 + e][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the two chunks issued past the end
    // ---- epilogue from registers: acc[2i + e][j][r] = channel nt*256 + i*32 + q*8 + e*4 + r of pixel pm0 + j*16 + l15
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
            if (m < p.M) *(u32x4 *)(p.y + m * p.Cout + nt * CK_BN + i * 32 + q * 8) = o;
        }
    }
}

template <int CIN>
int ck_launch(void *stream, const CKStreamK &p, int dtype)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    constexpr int CC = CIN / 64;
    constexpr int lds = (CK_S * CK_BN * 8 + (CK_BN / 4 > CC * 8 ? CK_BN / 4 : CC * 8)) * 16;
    static bool raised_d[USOT_MAX_DEV][2] = {};
    bool (&raised)[2] = raised_d[usot_dv];
    const void *fn = dtype ? (const void *)conv_kstream_kernel<CIN, true> : (const void *)conv_kstream_kernel<CIN, false>;
    if (!raised[dtype]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return USOT_ELAUNCH;
        raised[dtype] = true;
    }
    const long grid = (long)p.npanels * p.NT;
    if (grid > 0x7fffffffL) return USOT_EINVAL;
    if (dtype) hipLaunchKernelGGL((conv_kstream_kernel<CIN, true>), dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, p);
    else       hipLaunchKernelGGL((conv_kstream_kernel<CIN, false>), dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

}  // namespace

extern "C" int usot_conv_kstream_supported(int Cin, int Cout, int KH, int KW)
{
    return KH == 3 && KW == 3 && (Cin == 256 || Cin == 512) && Cout > 0 && Cout % 256 == 0;
}

/* y = act(conv3x3(x, w) + bias): NHWC dense, x / w / y in the storage type (dtype 0 = bf16, 1 = fp16), w [Cout][9 Cin] with
 * k = (kh*3 + kw)*Cin + ci, bias fp32 or NULL; stride 1 | 2, any pad / dilation; act USOT_ACT_NONE | USOT_ACT_RELU.
 * Shapes: usot_conv_kstream_supported(Cin, Cout, 3, 3) (Cin 256 | 512, Cout a multiple of 256). */
extern "C" int usot_conv_kstream_lp(void *stream, const void *x, const void *w, const float *bias, void *y,
                                    int N, int H, int W, int Cin, int Cout, int stride, int pad, int dil, int act, int dtype)
{
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || (dtype != 0 && dtype != 1) || !usot_conv_kstream_supported(Cin, Cout, 3, 3)) return USOT_EINVAL;
    if ((act != USOT_ACT_NONE && act != USOT_ACT_RELU) || stride < 1 || stride > 2 || pad < 0 || dil < 1 || pad > dil) return USOT_EINVAL;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)y) & 15) return USOT_EINVAL;
    CKStreamK p;
    p.x = (const uint16_t *)x; p.w = (const uint16_t *)w; p.bias = bias; p.y = (uint16_t *)y;
    p.N = N; p.H = H; p.W = W; p.stride = stride; p.pad = pad; p.dil = dil; p.act = act; p.Cout = Cout;
    p.OH = (H + 2 * pad - dil * 2 - 1) / stride + 1; p.OW = (W + 2 * pad - dil * 2 - 1) / stride + 1;
    if (p.OH <= 0 || p.OW <= 0) return USOT_EINVAL;
    // the centre tap of every output pixel lies inside the image (the kernel parks out-of-image taps there), offsets fit 32 bits
    if ((p.OH - 1) * stride - pad + dil >= H || (p.OW - 1) * stride - pad + dil >= W || (long)N * H * W * Cin * 2 >= 0xffffffffL) return USOT_EINVAL;
    const long M = (long)N * p.OH * p.OW;
    if (M > 0x7fffffffL) return USOT_EINVAL;
    p.M = (int)M; p.npanels = (int)((M + CK_BM - 1) / CK_BM); p.NT = Cout / CK_BN;
    return Cin == 512 ? ck_launch<512>(stream, p, dtype) : ck_launch<256>(stream, p, dtype);
}
