// Fused pair of pointwise (1x1) convolutions of the batched low-precision backbone (BASELINE config 3):
//
//     Y  = relu(T2 . W3^T + b3 + R)          a bottleneck's conv3 + BN + residual + ReLU (modules.py:48-56)
//     T  = act (Y  . W1^T + b1)              the NEXT block's conv1 + BN + ReLU (modules.py:40-42), or the
//                                            neck's 1x1 (connect.py:294-300, no activation)
//
// Unfused these are two HBM-bound launches: the wide map Y (C_out = 4 x C_mid) is written by the first and read
// back by the second, and each launch alternates a load phase, an MFMA phase and an HBM-heavy epilogue on a CU that
// holds ONE workgroup.  Here a persistent workgroup (8 wavefronts, one per CU) walks 64-pixel tiles:
//
//   GEMM1  acc[64 px x C_out] over K = C_mid.  T2 tile in LDS (LDS-DMA, prefetched during the previous tile's
//          GEMM2), W3 fragments straight from L2 into registers (a wave owns C_out / 8 channels, nothing to share).
//   epi 1  + bias + residual (prefetched into REGISTERS during the previous tile's GEMM2) -> ReLU -> bf16 -> global Y
//          and the LDS image of Y.  W3's rows are permuted on the host so that a lane's accumulator registers are
//          16 (8) CONTIGUOUS channels of one pixel: residual reads, Y stores and LDS writes are 16-byte pieces.
//   GEMM2  acc[64 px x C_next] over K = C_out, Y fragments from LDS, W1 fragments from L2.
//   epi 2  + bias, activation -> global T.
//
// Y never comes back from HBM (29 % fewer bytes per pair) and the next tile's HBM reads fly under GEMM2.
// Measured at batch 64 (scripts/lp_profile.py, pair fused vs the two launches): layer1 (64, 256, 64) 84 vs 115 us,
// (64, 256, 128) 95 vs 111, layer2 (128, 512, 128) 54 vs 64; (128, 512, 256) 70 vs 67 and layer3 (256, 1024, 256) 137 vs
// 137 — there a 64-pixel tile re-streams 1 MB of filters from L2 per tile (4x the unfused 256-pixel tiles), and with
// residual loads and Y stores ablated the pair still takes 90 us: L2 -> CU bandwidth, not HBM.  The engine fuses the
// first three shapes only (engine.FUSED_POINTWISE).
// LDS: 64 x C_mid x 2 B (T2) + 64 x C_out x 2 B (Y) = 160 KB for (256, 1024): the whole CU.
// Numerics: identical to the unfused kernels (fp32 accumulation in the same k order, one rounding of Y to the
// storage type, GEMM2 consumes the ROUNDED Y as the unfused conv1 does).
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

struct PwK {
    const uint16_t *t2, *w3p, *res, *w1;     // w3p / w1: FRAGMENT order (usot_pw_pair_layout)
    const float *b3, *b1;
    uint16_t *y, *t;
    int M, act2, ntiles;
};

__device__ __forceinline__ uint32_t pw_f2bf(float f)
{
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}
template <bool F16> __device__ __forceinline__ uint32_t pw_pack(float f)
{
    return F16 ? (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f) : pw_f2bf(f);
}
template <bool F16> __device__ __forceinline__ float pw_unpack(uint32_t h)
{
    return F16 ? (float)__builtin_bit_cast(_Float16, (uint16_t)h) : __builtin_bit_cast(float, h << 16);
}
template <bool F16> __device__ __forceinline__ f32x4 pw_mfma(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else               return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int CM, int CO, int CN, bool F16>
struct PwCfg {
    static constexpr int BM = 64;                  // pixels per tile
    static constexpr int CPW = CO / 8;             // GEMM1 channels per wave
    static constexpr int NT1 = CPW / 16;           // 16-channel MFMA tiles per wave
    static constexpr int PASSES = NT1 > 4 ? 2 : 1; // accumulators of one pass: 4 px-tiles x NTP x 4 registers
    static constexpr int NTP = NT1 / PASSES;
    static constexpr int SPAN = 4 * NTP;           // contiguous channels a lane owns per pass
    static constexpr int RPL = SPAN / 8;           // 16-byte pieces of that span
    static constexpr int KS1 = CM / 32, KS2 = CO / 32;
    static constexpr int CPR1 = CM / 8, CPR2 = CO / 8;      // 16-byte chunks per LDS row
    static constexpr int NT2 = CN / 16;
    static constexpr int TN2 = NT2 >= 8 ? NT2 / 8 : 1;      // GEMM2 n-tiles per wave
    static constexpr int NG2 = NT2 / TN2;                   // waves along n
    static constexpr int PXG = 8 / NG2;                     // waves along pixels
    static constexpr int TM2 = 4 / PXG;                     // px-tiles per wave
    static constexpr int LDS_BYTES = BM * CM * 2 + BM * CO * 2;
    static_assert(CM % 64 == 0 && CO % 256 == 0 && CN % 64 == 0, "channel counts");
    static_assert(NTP >= 2 && NTP <= 4 && RPL >= 1, "a lane owns 8 or 16 contiguous channels per pass");
    static_assert(NG2 * PXG == 8 && TM2 >= 1, "8 waves tile GEMM2");
};

// LDS swizzle of a row's 16-byte chunk index (ds_read_b128 lane groups, MI355X_MICROARCH.md): rows of >= 256 B XOR the
// low four chunk bits with row & 15; 128-byte rows (two per 256-byte bank line) XOR with (row >> 1) & 7.
template <int CPR> __device__ __forceinline__ int pw_swz(int row)
{
    return CPR >= 16 ? (row & 15) : ((row >> 1) & 7);
}

template <int CM, int CO, int CN, bool F16>
__global__ __launch_bounds__(512) void pw_pair_kernel(const PwK p)
{
    using G = PwCfg<CM, CO, CN, F16>;
    constexpr int BM = G::BM, NTP = G::NTP, PASSES = G::PASSES, SPAN = G::SPAN, RPL = G::RPL;
    extern __shared__ __attribute__((aligned(16))) u32x4 pw_smem[];
    u32x4 *sT2 = pw_smem;                              // [BM][CPR1]
    u32x4 *sY = pw_smem + BM * G::CPR1;                // [BM][CPR2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;

    // ---- T2 tile -> LDS by LDS-DMA: physical chunk g = instruction * 512 + tid; the swizzle sits on the SOURCE
    auto issue_t2 = [&](int tile) {
        const long bm0 = (long)tile * BM;
#pragma unroll
        for (int i = 0; i < BM * G::CPR1 / 512; ++i) {
            const int g = i * 512 + tid;
            const int row = g / G::CPR1, pc = g % G::CPR1;
            const int lc = pc ^ pw_swz<G::CPR1>(row);
            const long m = min(bm0 + row, (long)p.M - 1);
            // asm, not the builtin: beside a builtin LDS-DMA hipcc waits vmcnt(0) at every use of an ordinary load
            // (cdna_hip_programming.md, ".s-level traps" (b)), which would drain the W1 fragment ring of GEMM2 every
            // slice.  The compiler's own counted waits stay correct with these extra (older) entries in the queue.
            const uint16_t *src = p.t2 + m * CM + lc * 8;
            const uint32_t lds = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(sT2 + i * 512 + wave * 64));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
        }
    };
    // ---- residual of the lane's channel spans -> registers
    u32x4 rr[PASSES][4][RPL];
    const int chan_w = wave * G::CPW;                  // first GEMM1 channel of this wave
    auto load_res = [&](int tile) {
        const long bm0 = (long)tile * BM;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long m = min(bm0 + j * 16 + l15, (long)p.M - 1);
                const uint16_t *src = p.res + m * CO + chan_w + ps * (NTP * 16) + q * SPAN;
#pragma unroll
                for (int k = 0; k < RPL; ++k) rr[ps][j][k] = *(const u32x4 *)(src + k * 8);
            }
    };

    int tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    issue_t2(tile);
    load_res(tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA is invisible to the compiler's wait counts
    __syncthreads();

    for (;;) {
        const long bm0 = (long)tile * BM;
        // ================= GEMM1 + epilogue 1, one or two channel passes =================
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            f32x4 acc[NTP][4];
#pragma unroll
            for (int t = 0; t < NTP; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // fragment-packed filters: [wave][pass][k-slice][tile][lane][8] — every fragment load is one contiguous KiB
            // (row-strided fragment loads, 16 half-lines per instruction, made the loop texture-addresser-bound)
            const uint16_t *wfr = p.w3p + ((long)(wave * PASSES + ps) * G::KS1 * NTP * 64 + lane) * 8;
            // two k-slices per iteration, W3 fragments double-buffered by name (static register indices); the loop
            // itself is NOT unrolled: hipcc otherwise hoists every slice's fragment loads and spills
            u32x4 wa[NTP], wb[NTP];
#pragma unroll
            for (int t = 0; t < NTP; ++t) wa[t] = *(const u32x4 *)(wfr + t * 512);
            auto slice = [&](int sl, const u32x4 (&w)[NTP]) {
                u32x4 xf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = j * 16 + l15;
                    xf[j] = sT2[row * G::CPR1 + ((sl * 4 + q) ^ pw_swz<G::CPR1>(row))];
                }
#pragma unroll
                for (int t = 0; t < NTP; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][j] = pw_mfma<F16>(w[t], xf[j], acc[t][j]);
            };
#pragma unroll 1
            for (int s = 0; s < G::KS1; s += 2) {
#pragma unroll
                for (int t = 0; t < NTP; ++t) wb[t] = *(const u32x4 *)(wfr + ((s + 1) * NTP + t) * 512);
                slice(s, wa);
                const int s2 = s + 2 < G::KS1 ? s + 2 : s;          // the last prefetch re-reads a resident line
#pragma unroll
                for (int t = 0; t < NTP; ++t) wa[t] = *(const u32x4 *)(wfr + (s2 * NTP + t) * 512);
                slice(s + 1, wb);
            }
            // epilogue 1: the lane owns channels [c0, c0 + SPAN) of pixel row j*16 + l15 for every j
            const int c0 = chan_w + ps * (NTP * 16) + q * SPAN;
            f32x4 bias[NTP];
#pragma unroll
            for (int t = 0; t < NTP; ++t) bias[t] = *(const f32x4 *)(p.b3 + c0 + t * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = j * 16 + l15;
                const long m = bm0 + row;
                uint32_t h[SPAN];
#pragma unroll
                for (int t = 0; t < NTP; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = t * 4 + r;                         // channel c0 + e
                        const uint32_t rw = rr[ps][j][e / 8][(e % 8) / 2];
                        const float rv = pw_unpack<F16>((e & 1) ? (rw >> 16) : (rw & 0xffffu));
                        h[e] = pw_pack<F16>(fmaxf(acc[t][j][r] + bias[t][r] + rv, 0.0f));
                    }
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = h[k * 8 + 2 * e] | (h[k * 8 + 2 * e + 1] << 16);
                    sY[row * G::CPR2 + ((c0 / 8 + k) ^ pw_swz<G::CPR2>(row))] = o;
                    if (m < p.M) *(u32x4 *)(p.y + m * CO + c0 + k * 8) = o;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's Y image is in LDS ...
        __builtin_amdgcn_s_barrier();                        // ... and everyone's; T2 is free again (raw barrier: the Y
        asm volatile("" ::: "memory");                       // stores above stay in flight)

        // ================= prefetch the next tile under GEMM2 =================
        const int next = tile + gridDim.x;
        if (next < p.ntiles) {
            issue_t2(next);
            load_res(next);
        }

        // ================= GEMM2 + epilogue 2 =================
        {
            const int ng = wave % G::NG2, pxg = wave / G::NG2;
            f32x4 acc[G::TN2][G::TM2];
#pragma unroll
            for (int a = 0; a < G::TN2; ++a)
#pragma unroll
                for (int b = 0; b < G::TM2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            const uint16_t *wfr = p.w1 + ((long)ng * G::KS2 * G::TN2 * 64 + lane) * 8;      // [n-group][k-slice][tile][lane][8]
            // W1 fragments: four k-slices in flight (ring of named buffers), loop not unrolled beyond the ring
            constexpr int PF = 4;
            static_assert(G::KS2 % PF == 0, "ring");
            u32x4 wf[PF][G::TN2];
#pragma unroll
            for (int d = 0; d < PF; ++d)
#pragma unroll
                for (int a = 0; a < G::TN2; ++a) wf[d][a] = *(const u32x4 *)(wfr + (d * G::TN2 + a) * 512);
#pragma unroll 1
            for (int s0 = 0; s0 < G::KS2; s0 += PF) {
#pragma unroll
                for (int d = 0; d < PF; ++d) {
                    const int sl = s0 + d;
                    u32x4 xf[G::TM2];
#pragma unroll
                    for (int b = 0; b < G::TM2; ++b) {
                        const int row = (pxg * G::TM2 + b) * 16 + l15;
                        xf[b] = sY[row * G::CPR2 + ((sl * 4 + q) ^ pw_swz<G::CPR2>(row))];
                    }
#pragma unroll
                    for (int a = 0; a < G::TN2; ++a)
#pragma unroll
                        for (int b = 0; b < G::TM2; ++b) acc[a][b] = pw_mfma<F16>(wf[d][a], xf[b], acc[a][b]);
                    const int sn = sl + PF < G::KS2 ? sl + PF : sl;   // refill this ring slot (tail: a resident line)
#pragma unroll
                    for (int a = 0; a < G::TN2; ++a) wf[d][a] = *(const u32x4 *)(wfr + (sn * G::TN2 + a) * 512);
                }
            }
#pragma unroll
            for (int a = 0; a < G::TN2; ++a) {
                const int n0 = (ng * G::TN2 + a) * 16 + q * 4;
                const f32x4 bias = *(const f32x4 *)(p.b1 + n0);
#pragma unroll
                for (int b = 0; b < G::TM2; ++b) {
                    const long m = bm0 + (pxg * G::TM2 + b) * 16 + l15;
                    if (m >= p.M) continue;
                    f32x4 v = acc[a][b] + bias;
                    if (p.act2 == USOT_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                    u32x2 o;
                    o[0] = usot_pack2_lp<F16>(v[0], v[1]);
                    o[1] = usot_pack2_lp<F16>(v[2], v[3]);
                    *(u32x2 *)(p.t + m * CN + n0) = o;
                }
            }
        }
        if (next >= p.ntiles) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next T2 tile has landed
        __syncthreads();                                     // ... for every wave, and all are done reading this tile's Y image
        tile = next;
    }
}

template <int CM, int CO, int CN>
int pw_launch(hipStream_t s, const PwK &p, int dtype)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    using G16 = PwCfg<CM, CO, CN, true>;
    constexpr int lds = G16::LDS_BYTES;
    static bool raised_d[USOT_MAX_DEV][2] = {};
    bool (&raised)[2] = raised_d[usot_dv];
    const void *fn = dtype ? (const void *)pw_pair_kernel<CM, CO, CN, true> : (const void *)pw_pair_kernel<CM, CO, CN, false>;
    if (lds > 64 * 1024 && !raised[dtype]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return USOT_ELAUNCH;
        raised[dtype] = true;
    }
    int cus = 256;
    {
        static int cached_d[USOT_MAX_DEV] = {};
    int &cached = cached_d[usot_dv];
        if (!cached) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                cached = prop.multiProcessorCount;
            else
                cached = 256;
        }
        cus = cached;
    }
    const int wgs_per_cu = lds > 48 * 1024 ? 1 : 2;          // residency: LDS, and 128 VGPRs per lane at two 8-wave workgroups
    const int grid = p.ntiles < cus * wgs_per_cu ? p.ntiles : cus * wgs_per_cu;
    if (dtype) hipLaunchKernelGGL((pw_pair_kernel<CM, CO, CN, true>), dim3(grid), dim3(512), lds, s, p);
    else       hipLaunchKernelGGL((pw_pair_kernel<CM, CO, CN, false>), dim3(grid), dim3(512), lds, s, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

}  // namespace

extern "C" int usot_pw_pair_supported(int CM, int CO, int CN);

/* Fragment order of the two filter banks handed to usot_pw_pair_lp.  For `which` = 0 (conv3 bank w3 [CO][CM]) or
 * 1 (next conv's bank w1 [CN][CO]) fills, for every 16-byte chunk c of the packed buffer (8 consecutive elements),
 * row[c] and k0[c]: packed[c*8 + e] = w[row[c]][k0[c] + e].  Returns the number of chunks (rows x cols / 8).
 * conv3: [wave 8][pass][k-slice][16-row tile][lane 64]; the rows a lane's accumulators stand for are permuted so that
 * they are contiguous output channels (see the kernel header).  next conv: [n-group][k-slice][tile][lane 64]. */
extern "C" int usot_pw_pair_layout(int CM, int CO, int CN, int which, int32_t *row, int32_t *k0)
{
    if (!usot_pw_pair_supported(CM, CO, CN) || !row || !k0) return USOT_EINVAL;
    int c = 0;
    if (which == 0) {
        const int cpw = CO / 8, nt1 = cpw / 16, passes = nt1 > 4 ? 2 : 1, ntp = nt1 / passes, ks1 = CM / 32;
        for (int w = 0; w < 8; ++w)
            for (int ps = 0; ps < passes; ++ps)
                for (int s = 0; s < ks1; ++s)
                    for (int t = 0; t < ntp; ++t)
                        for (int lane = 0; lane < 64; ++lane, ++c) {
                            const int rho = lane & 15, q = lane >> 4;       // MFMA A row, k-chunk
                            row[c] = w * cpw + ps * ntp * 16 + (rho >> 2) * 4 * ntp + t * 4 + (rho & 3);
                            k0[c] = s * 32 + q * 8;
                        }
        return c;
    }
    if (which == 1) {
        const int nt2 = CN / 16, tn2 = nt2 >= 8 ? nt2 / 8 : 1, ng2 = nt2 / tn2, ks2 = CO / 32;
        for (int ng = 0; ng < ng2; ++ng)
            for (int s = 0; s < ks2; ++s)
                for (int a = 0; a < tn2; ++a)
                    for (int lane = 0; lane < 64; ++lane, ++c) {
                        row[c] = (ng * tn2 + a) * 16 + (lane & 15);
                        k0[c] = s * 32 + (lane >> 4) * 8;
                    }
        return c;
    }
    return USOT_EINVAL;
}

extern "C" int usot_pw_pair_supported(int CM, int CO, int CN)
{
    return (CM == 256 && CO == 1024 && CN == 256) || (CM == 128 && CO == 512 && CN == 128) || (CM == 64 && CO == 256 && CN == 64) ||
           (CM == 64 && CO == 256 && CN == 128) || (CM == 128 && CO == 512 && CN == 256);
}

extern "C" int usot_pw_pair_lp(void *stream, const usot_pw_pair_desc *d, int dtype)
{
    if (!d || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    if (!d->t2 || !d->w3p || !d->b3 || !d->res || !d->y || !d->w1 || !d->b1 || !d->t || d->M <= 0) return USOT_EINVAL;
    if (d->act2 != USOT_ACT_NONE && d->act2 != USOT_ACT_RELU) return USOT_EINVAL;
    const void *ptrs[] = {d->t2, d->w3p, d->b3, d->res, d->y, d->w1, d->b1, d->t};
    for (const void *q : ptrs)
        if ((uintptr_t)q % 16) return USOT_EINVAL;
    PwK p;
    p.t2 = (const uint16_t *)d->t2; p.w3p = (const uint16_t *)d->w3p; p.res = (const uint16_t *)d->res; p.w1 = (const uint16_t *)d->w1;
    p.b3 = d->b3; p.b1 = d->b1; p.y = (uint16_t *)d->y; p.t = (uint16_t *)d->t;
    p.M = d->M; p.act2 = d->act2; p.ntiles = (d->M + 63) / 64;
    hipStream_t s = (hipStream_t)stream;
    if (d->CM == 256 && d->CO == 1024 && d->CN == 256) return pw_launch<256, 1024, 256>(s, p, dtype);
    if (d->CM == 128 && d->CO == 512 && d->CN == 128) return pw_launch<128, 512, 128>(s, p, dtype);
    if (d->CM == 128 && d->CO == 512 && d->CN == 256) return pw_launch<128, 512, 256>(s, p, dtype);
    if (d->CM == 64 && d->CO == 256 && d->CN == 64) return pw_launch<64, 256, 64>(s, p, dtype);
    if (d->CM == 64 && d->CO == 256 && d->CN == 128) return pw_launch<64, 256, 128>(s, p, dtype);
    return USOT_EINVAL;
}
