// A layer3 bottleneck's conv2 -> conv3 of the batched low-precision backbone in ONE launch (BASELINE config 3;
// modules.py:43-56), the fusion VERDICT r3 / r4 asked to be BUILT and measured rather than priced:
//
//     T2[M][256]  = relu(conv3x3(T1; w2) + b2)            256 -> 256, stride 1, any pad / dilation (layer3: 2 / 2)
//     Y [M][1024] = relu(T2 . w3^T + b3 + R)              the 1x1 expansion + BN + residual + ReLU
//
// As two launches T2 (31.5 MB at batch 64) is written by the 256 x 256 implicit-GEMM tile and read back by the
// pixel-stationary panel kernel (csrc/pw_panel.hip): 63 of the pair's 346 MB.  Here a workgroup (16 wavefronts) owns a
// PANEL of 256 pixels through both convolutions:
//   phase 1  conv2's k-loop exactly as tile 32 of csrc/conv_bf16.hip (256 x 256 x 64 k-tiles, two LDS stages filled by
//            LDS-DMA, one raw barrier per k-tile, a wave = 64 pixels x 64 channels of accumulators);
//   phase 2  bias + ReLU + rounding to the storage type, the T2 panel goes to LDS (128 KB: the two stages are free), rows
//            swizzled for the reads of
//   phase 3  every wave takes the B fragments of ITS 16 pixels (all 256 k: 32 registers) out of LDS - after the barrier
//            that follows the whole LDS is free again;
//   phase 4  the panel kernel's loop: w3 streams through a ring of three 32 KB slabs (64 channels x 256 k, rows permuted
//            so that a lane's accumulators are two runs of 8 contiguous channels), leaders / trailers half an interval
//            apart, residual prefetched one group ahead, register epilogue in 16-byte pieces.
// Same arithmetic in the same order as the two launches: the result is BIT-IDENTICAL to tile 32 followed by
// usot_pw_panel_lp (tests/test_gpu_ops.py::test_conv_pw_fused_equals_the_two_launches).
// What it cannot do (and why the gain is bounded): a launch is ONE round of 241 workgroups, one per CU, so
// the matrix-pipe phase (1) and the HBM phase (4: 252 MB of residual + Y) of every CU run one after the other; the
// fusion removes T2's round trip and a launch boundary, it does not overlap the two phases.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "usot_hip.h"
#include "common.h"

extern "C" int usot_conv_pw_pixels(int64_t M);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct CpK {
    const uint16_t *x, *w2, *w3, *res, *zero, *w1;
    const float *b2, *b3, *b1;
    uint16_t *y, *t;
    int H, W, OW, KW, stride, pad_h, pad_w, dil_h, dil_w;
    int M, P, KT, npanels, act2;
    int rs;                        // phase 1 on the row-shared k-loop (3 x 3, stride 1, pad = dil <= 4)
};

__device__ __attribute__((aligned(16))) uint32_t cp_zero16[4] = {0u, 0u, 0u, 0u};     // source of padding taps (LDS-DMA)

#ifndef CP_WD
#define CP_WD 2
#endif
constexpr int CP_LDC = 8;                                  // 16-byte chunks per staged row (64 elements)

// NW wavefronts own a panel of 16 NW pixels: 16 (256 pixels) or 8 (128 pixels: batches whose 256-pixel panels would leave CUs
// idle).  CM = conv2's width (= its input's: 256 in layer3, 128 in layer2), conv3 expands to 4 CM; CN > 0: the pair form -
// the NEXT block's conv1 (4 CM -> CN) rides along as in csrc/pw_panel.hip (Y's sixteen channels per lane, rounded, are the B
// fragments of the second GEMM; its bank's k-slice of the group sits in the same slab ring).
template <int NW, int CM, int CN>
struct CpCfg {
    static constexpr int BM = NW * 16, NTHR = NW * 64;
    static constexpr int CIN = CM, CO = 4 * CM;
    static constexpr int CCH = CIN / 64;                   // channel chunks per tap
    static constexpr int CPR = CM / 8;                     // 16-byte chunks per T2 row / per w3 slab row
    static constexpr int SLAB0 = 64 * CPR;                 // chunks of a w3 slab (64 channels x CM k)
    // layer3's next conv1 (1024 -> 256) has no register pair form (64 more accumulators than a 16-wave workgroup has): CM = 256 with
    // CN > 0 is the PHASE-5 form instead - after conv3 the workgroup reads its own Y panel back (L2 / Infinity Cache: it has just
    // written it) and runs conv1 as a 16-k-tile implicit GEMM on the freed LDS
    static constexpr bool P5 = CN == 256;                  // (layer3: CM = 256; layer2's last block: CM = 128, next conv1 512 -> 256)
    static constexpr int CNR = P5 ? 0 : CN;                // width of the pair form's second GEMM
    static constexpr int SLAB1 = CNR * 8;                  // ... of w1's slab (CN channels x this group's 64 k)
    static constexpr int SLAB = SLAB0 + SLAB1;
    static constexpr int G = CO / 64;                      // channel groups of conv3
    static constexpr int KS = CM / 32;                     // MFMA k-steps of conv3
    static constexpr int NB1 = CNR / 16;                   // 16-channel blocks of T (pair form)
    static constexpr int TM = 4, TN = CM / 64;             // a wave = 64 pixels x CM / 4 channels of conv2
    static constexpr int WM = BM / (TM * 16);              // waves along the pixels (4 | 2); along the channels: 4
    static constexpr int RPP = NTHR / 8;                   // tile rows one staging pass covers
    static constexpr int XI = BM / RPP, WI = CM / RPP;     // DMA instructions per lane, k-tile and operand
    // row-shared form of phase 1 (below): an activation stage holds XROWS = BM + 2 x 4 halo rows and serves the three kw taps of a
    // (kh, channel chunk) group; + one row of zeros for taps outside the image
    static constexpr int XROWS = BM + 8, NBLK = XROWS / 8, PERK = (NBLK + 2) / 3;
    static constexpr int STAGES = 2 * (XROWS + CM) * CP_LDC * 16 + 128, RING = 3 * SLAB * 16 + (CO + CN) * 4;
    static constexpr int STAGES5 = P5 ? (3 * BM + 2 * CN) * CP_LDC * 16 : 0;       // phase 5: three Y stages + two filter stages
    static constexpr int LDS = STAGES5 > (STAGES > RING ? STAGES : RING) ? STAGES5 : (STAGES > RING ? STAGES : RING);
    static_assert(PERK <= NW, "a k-tile's share of the next activation stage: at most one DMA instruction per wave");
    static_assert(NW == 16 || NW == 8, "panels of 256 or 128 pixels");
    static_assert(CM == 256 || CM == 128, "layer3 / layer2 widths");
    static constexpr int TN5 = CN / 64, WI5 = CN / RPP > 0 ? CN / RPP : 1;      // phase 5: a wave = 64 pixels x CN / 4 channels
    static_assert(CN % 64 == 0 && WI >= 1 && CM % RPP == 0 && (!P5 || CN % RPP == 0), "shape");
    static_assert(STAGES >= BM * CPR * 16, "the T2 panel fits conv2's stages");
    static_assert(SLAB0 % NTHR == 0 && SLAB1 % NTHR == 0, "whole DMA instructions per slab");
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <bool F16> __device__ __forceinline__ f32x4 cp_mfma(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else               return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ float cp_unpack(uint32_t h)
{
    return F16 ? (float)__builtin_bit_cast(_Float16, (uint16_t)h) : __builtin_bit_cast(float, h << 16);
}
__device__ __forceinline__ int cp_swz(int row) { return (row >> 1) & 7; }     // staged k-tile rows (conv_bf16.hip)

__device__ __forceinline__ int cp_xcd_remap(int b, int total)                  // consecutive panels (shared halo rows) on one XCD
{
    const int q = total >> 3, r = total & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void cp_wait_vm(int n)          // s_waitcnt vmcnt(n), n wave-uniform at run time
{
    switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// raw workgroup barrier (leaders and trailers run different instruction streams in phase 4: csrc/pw_panel.hip)
__device__ __forceinline__ void cp_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <bool F16, int NW, int CM, int CN, bool RS>
__global__ __launch_bounds__(NW * 64) void conv_pw_kernel(const CpK p)
{
    using Cf = CpCfg<NW, CM, CN>;
    constexpr int BM = Cf::BM, LDC = CP_LDC, NTHR = Cf::NTHR, RPP = Cf::RPP, XI = Cf::XI, WI = Cf::WI, TM = Cf::TM, TN = Cf::TN, WM = Cf::WM;
    constexpr int CP_CIN = Cf::CIN, CP_CCH = Cf::CCH, CP_CPR = Cf::CPR, CP_SLAB = Cf::SLAB, CP_G = Cf::G, CP_KS = Cf::KS, CP_CO = Cf::CO;
    constexpr int NB1 = Cf::NB1;
    extern __shared__ __attribute__((aligned(16))) u32x4 cp_smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, quad = lane >> 4;
    const int panel = cp_xcd_remap(blockIdx.x, p.npanels);
    const int bm0 = panel * BM;

    // asm LDS-DMA with a plain 32-bit LDS address: invisible to the compiler's waitcnt bookkeeping, completion counted by hand
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)cp_smem;
    auto dma16u = [&](const uint16_t *src, uint32_t lds) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
    };

    // ------------------------------------------------------------------ phase 1: conv2, BM pixels x CM channels
    f32x4 acc2[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wm = wave % WM, wn = wave / WM;
    if constexpr (RS) {
        // ---- ROW-SHARED k-loop (3 x 3, stride 1, "same" padding: pad = dil <= 4, OW = W).  In the flattened pixel index the
        // tap (kh, kw) of output pixel m is the tap (kh, centre) of pixel m + (kw - 1) dil whenever that pixel is in the same
        // image row - so ONE staged tile per (kh, channel chunk), BM + 8 rows (pixels bm0 - 4 ... bm0 + BM + 3 at their kh row),
        // serves the three kw k-tiles: a lane reads its B fragments (kw - 1) dil rows up or down, or from a row of zeros when the
        // tap leaves the image row.  43 instead of 64 LDS-DMA instructions per k-tile and CU at 256 x 256 (a k-tile's DMA issue is
        // ~ 13 cycles per instruction next to its 2 048 MFMA cycles, csrc/conv_bf16.hip).  k order: (kh, chunk, kw).
        constexpr int XROWS = Cf::XROWS, NBLK = Cf::NBLK, PERK = Cf::PERK, XST = XROWS * LDC;
        const u32x4 *sW = cp_smem + 2 * XST;
        const int zchunk = 2 * XST + 2 * CM * LDC;         // the row of zeros, in 16-byte chunks from cp_smem
        if (tid < 8) cp_smem[zchunk + tid] = u32x4{0u, 0u, 0u, 0u};
        // this lane's rows of an activation stage: block kw * PERK + wave (8 rows) in the k-tile kw of a group
        // (32-bit element offsets: the launcher checks the map has fewer than 2^31 elements; a row that does not exist is marked
        //  by an oh far outside the image so that every kh fails the range check)
        uint32_t xs_off[3];
        int xs_oh[3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int row = (kw * PERK + wave) * 8 + (lane >> 3);
            const int v = bm0 - 4 + row;
            const bool ok = wave < PERK && kw * PERK + wave < NBLK && v >= 0 && v < p.M;
            const int vv = ok ? v : 0;
            const int n = vv / p.P, pix = vv - n * p.P;
            const int oh = pix / p.OW, ow = pix - oh * p.OW;
            xs_oh[kw] = ok ? oh - p.pad_h : -(1 << 20);
            xs_off[kw] = (uint32_t)((n * p.H * p.W + ow) * CP_CIN + ((lane & 7) ^ cp_swz(row)) * 8);
        }
        auto issue_x = [&](auto kwc, int g, int stage) {
            constexpr int kw = decltype(kwc)::value;
            if (wave < PERK && kw * PERK + wave < NBLK) {             // wave-uniform
                const int kh = g / CP_CCH, cc = g - kh * CP_CCH;
                const int ih = xs_oh[kw] + kh * p.dil_h;
                const bool ok = (unsigned)ih < (unsigned)p.H;
                dma16u(ok ? p.x + (xs_off[kw] + (uint32_t)(ih * p.W * CP_CIN + cc * 64)) : p.zero,
                       __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((stage * XST + (kw * PERK + wave) * 64) * 16)));
            }
        };
        const int lr = tid >> 3;
        const uint32_t K2 = (uint32_t)p.KT * 64u;
        const uint32_t wbase = (uint32_t)lr * K2 + (uint32_t)(((tid & 7) ^ cp_swz(lr)) * 8);     // swz(lr + RPP i) = swz(lr)
        const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((2 * XST + wave * 8 * LDC) * 16));
        auto issue_w = [&](int g, int kw, int buf) {
            const int kh = g / CP_CCH, cc = g - kh * CP_CCH;
            const uint32_t koff = (uint32_t)(((kh * 3 + kw) * CP_CCH + cc) * 64);
#pragma unroll
            for (int i = 0; i < WI; ++i)
                dma16u(p.w2 + (wbase + koff + (uint32_t)(RPP * i) * K2), ldsw + (uint32_t)((buf * CM + RPP * i) * LDC * 16));
        };
        // this lane's pixels (one per 16-pixel block of the wave): may the left / right tap be read (same image row)?  bit j: left
        // tap of block j, bit 4 + j: right tap
        uint32_t tapmask = 0;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int ow = (bm0 + wm * TM * 16 + j * 16 + l15) % p.OW;
            tapmask |= (ow - p.dil_w >= 0 ? 1u : 0u) << j;
            tapmask |= (ow + p.dil_w < p.W ? 1u : 0u) << (4 + j);
        }
        const int row0 = 4 + wm * TM * 16 + l15;
        const int ng = 3 * CP_CCH;
        using K0 = std::integral_constant<int, 0>;
        using K1 = std::integral_constant<int, 1>;
        using K2c = std::integral_constant<int, 2>;
        issue_x(K0{}, 0, 0); issue_x(K1{}, 0, 0); issue_x(K2c{}, 0, 0);
        issue_w(0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        auto ktile = [&](auto kwc, int g) {
            constexpr int kw = decltype(kwc)::value;
            const int t = g * 3 + kw;
            // the next k-tile's filters, this k-tile's share of the next group's activation stage
            if (kw < 2) issue_w(g, kw + 1, (t + 1) & 1);
            else if (g + 1 < ng) issue_w(g + 1, 0, (t + 1) & 1);
            if (g + 1 < ng) issue_x(kwc, g + 1, (g + 1) & 1);
            int sh = (kw - 1) * p.dil_w;
            asm volatile("" : "+s"(sh));                 // recomputed per k-tile: hoisted, the 12 (kw, block) fragment addresses of
                                                           // both stages would live in registers across the loop (and spill)
            int a0[TM];
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int row = row0 + j * 16 + sh;
                const bool in = kw == 1 || ((tapmask >> ((kw == 0 ? 0 : 4) + j)) & 1u);
                a0[j] = in ? (g & 1) * XST + row * LDC + (quad ^ cp_swz(row)) : zchunk + quad;
            }
            const u32x4 *cW = sW + ((t & 1) * CM + wn * TN * 16 + l15) * LDC;
            const int sq = cp_swz(l15);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 wf[TN], xf[TM];
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[i] = cW[i * 16 * LDC + ((ks * 4 + quad) ^ sq)];
#pragma unroll
                for (int j = 0; j < TM; ++j) xf[j] = cp_smem[a0[j] ^ (ks * 4)];      // chunk (4 ks + quad) ^ swz = first ^ 4 ks
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc2[i][j] = cp_mfma<F16>(wf[i], xf[j], acc2[i][j]);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        for (int g = 0; g < ng; ++g) {
            ktile(K0{}, g);
            ktile(K1{}, g);
            ktile(K2c{}, g);
        }
    } else {
        const int lr = tid >> 3;
        const int kc = (tid & 7) ^ cp_swz(lr);             // the swizzle sits on the SOURCE address (the DMA's LDS position is fixed)
        int x_ih0[XI], x_iw0[XI];
        long x_nb[XI];
        bool x_ok[XI];
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int m = bm0 + lr + RPP * i;
            x_ok[i] = m < p.M;
            const int mm = x_ok[i] ? m : 0;
            const int n = mm / p.P, pix = mm - n * p.P;
            const int oh = pix / p.OW, ow = pix - oh * p.OW;
            x_ih0[i] = oh * p.stride - p.pad_h;
            x_iw0[i] = ow * p.stride - p.pad_w;
            x_nb[i] = (long)n * p.H * p.W * CP_CIN + kc * 8;
        }
        const long K2 = (long)p.KT * 64;
        const uint16_t *wp[WI];
#pragma unroll
        for (int i = 0; i < WI; ++i) wp[i] = p.w2 + (long)(lr + RPP * i) * K2 + kc * 8;
        const uint16_t *xp[XI];
        bool xin[XI];
        int cur_tap = 0, cur_cc = 0;
        auto set_tap = [&](int tap) {
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int ih = x_ih0[i] + kh * p.dil_h, iw = x_iw0[i] + kw * p.dil_w;
                xin[i] = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                xp[i] = p.x + x_nb[i] + ((long)ih * p.W + iw) * CP_CIN;
            }
        };
        set_tap(0);
        const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(wave * 8 * LDC * 16));
        auto issue_tile = [&](int buf, bool advance) {
            const int c0 = cur_cc * 64;
            const uint32_t bx = ldsw + (uint32_t)(buf * BM * LDC * 16), bw = ldsw + (uint32_t)((2 * BM + buf * CM) * LDC * 16);
#pragma unroll
            for (int i = 0; i < XI; ++i) dma16u(xin[i] ? xp[i] + c0 : p.zero, bx + (uint32_t)(RPP * i * LDC * 16));
#pragma unroll
            for (int i = 0; i < WI; ++i) {
                dma16u(wp[i], bw + (uint32_t)(RPP * i * LDC * 16));
                wp[i] += advance ? 64 : 0;
            }
            if (advance && ++cur_cc == CP_CCH) {
                cur_cc = 0;
                set_tap(++cur_tap);
            }
        };
        const u32x4 *sX = cp_smem, *sW = cp_smem + 2 * BM * LDC;
        const int sq = cp_swz(l15);                        // rows differ from l15 by multiples of 16
        const int nt = p.KT;
        issue_tile(0, nt > 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt) issue_tile(cur ^ 1, t + 2 < nt);
            const u32x4 *cX = sX + (cur * BM + wm * TM * 16 + l15) * LDC;
            const u32x4 *cW = sW + (cur * CM + wn * TN * 16 + l15) * LDC;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 wf[TN], xf[TM];
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[i] = cW[i * 16 * LDC + ((ks * 4 + quad) ^ sq)];
#pragma unroll
                for (int j = 0; j < TM; ++j) xf[j] = cX[j * 16 * LDC + ((ks * 4 + quad) ^ sq)];
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc2[i][j] = cp_mfma<F16>(wf[i], xf[j], acc2[i][j]);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    {
        // -------------------------------------------------------------- phase 2: T2 = relu(acc + b2) -> LDS, storage type
        // row r of the panel = CP_CPR chunks of 8 channels, chunk c at position c ^ (r & 15): a wave's fragment reads below
        // (16 rows x 4 consecutive chunks per instruction) then fall into 16 distinct 16-byte slots per lane group
        char *sT = (char *)cp_smem;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int ch = wn * TN * 16 + i * 16 + quad * 4;
            const f32x4 b = *(const f32x4 *)(p.b2 + ch);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int row = wm * TM * 16 + j * 16 + l15;
                f32x4 v = acc2[i][j] + b;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                u32x2 o;
                o[0] = usot_pack2_lp<F16>(v[0], v[1]);
                o[1] = usot_pack2_lp<F16>(v[2], v[3]);
                *(u32x2 *)(sT + ((row * CP_CPR + ((ch >> 3) ^ (row & 15))) * 16 + ((ch >> 2) & 1) * 8)) = o;
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ phase 3: this wave's 16 pixels, all of k, into registers
    u32x4 xf[CP_KS];
#pragma unroll
    for (int ks = 0; ks < CP_KS; ++ks) xf[ks] = cp_smem[(wave * 16 + l15) * CP_CPR + ((ks * 4 + quad) ^ l15)];
    __syncthreads();                                       // the T2 panel has been read out: the LDS is free

    // ------------------------------------------------------------------ phase 4: conv3 over the panel (csrc/pw_panel.hip, PB = 1)
    float *sBias = (float *)(cp_smem + 3 * CP_SLAB);
    for (int i = tid; i < CP_CO; i += NTHR) sBias[i] = p.b3[i];
    if constexpr (NB1 > 0)
        for (int i = tid; i < CN; i += NTHR) sBias[CP_CO + i] = p.b1[i];
    // slab of group g -> ring slot: physical chunk c = i * NTHR + tid holds logical chunk pc ^ (row & 15) of slab row `row`,
    // row = MFMA row rho of 16-channel block blk = channel g * 64 + (blk >> 1) * 32 + (rho >> 2) * 8 + (blk & 1) * 4 + (rho & 3)
    auto issue_slab = [&](int g, int slot) {
        const uint32_t base = lds0 + (uint32_t)(slot * CP_SLAB * 16) + (uint32_t)(wave * 64 * 16);
#pragma unroll
        for (int i = 0; i < Cf::SLAB0 / NTHR; ++i) {
            const int c = i * NTHR + tid;
            const int row = c / CP_CPR, pc = c % CP_CPR;
            const int lc = pc ^ (row & 15);
            const int blk = row >> 4, rho = row & 15;
            const int ch = g * 64 + (blk >> 1) * 32 + (rho >> 2) * 8 + (blk & 1) * 4 + (rho & 3);
            dma16u(p.w3 + (long)ch * CM + lc * 8, __builtin_amdgcn_readfirstlane(base + (uint32_t)(i * NTHR * 16)));
        }
        if constexpr (NB1 > 0) {
            // w1's k-slice of the group: row = T channel (blk >> 2) * 64 + ((blk >> 1) & 1) * 32 + (rho >> 2) * 8 + (blk & 1) * 4 + (rho & 3)
            // (blk = 16-channel block of T), 8 chunks of 8 k, chunk pc holds logical chunk pc ^ ((row >> 1) & 7)
#pragma unroll
            for (int i = 0; i < Cf::SLAB1 / NTHR; ++i) {
                const int c = i * NTHR + tid;
                const int row = c / 8, pc = c % 8;
                const int lc = pc ^ ((row >> 1) & 7);
                const int blk = row >> 4, rho = row & 15;
                const int ch = (blk >> 2) * 64 + ((blk >> 1) & 1) * 32 + (rho >> 2) * 8 + (blk & 1) * 4 + (rho & 3);
                dma16u(p.w1 + (long)ch * CP_CO + g * 64 + lc * 8,
                       __builtin_amdgcn_readfirstlane(base + (uint32_t)((Cf::SLAB0 + i * NTHR) * 16)));
            }
        }
    };
    const long pm = (long)bm0 + wave * 16 + l15;           // this lane's pixel
    const long pmc = pm < (long)p.M ? pm : (long)p.M - 1;
    const bool full = (long)bm0 + wave * 16 + 16 <= (long)p.M;     // wave-uniform: every pixel row of this wave exists
    const int q = quad;

    auto run = [&](auto role) {
        constexpr bool trail = decltype(role)::value;      // two fully specialised instruction streams
        issue_slab(0, 0);
        // residual of the lane's two 8-channel runs: ONE register set (128 registers per wave at 16 waves), loaded right after the
        // epilogue that consumed the previous group's - a whole GEMM phase ahead of its own epilogue
        u32x4 rr[2];
        auto load_res = [&](int g) {
            const int c0 = (g < CP_G ? g : CP_G - 1) * 64 + q * 8;   // past the last group: a harmless re-read keeps the op count fixed
            rr[0] = *(const u32x4 *)(p.res + pmc * CP_CO + c0);
            rr[1] = *(const u32x4 *)(p.res + pmc * CP_CO + c0 + 32);
        };
        load_res(0);
        f32x4 acc[4];
        f32x4 acct[NB1 ? NB1 : 1];                         // the second GEMM's accumulators live across the groups
#pragma unroll
        for (int n = 0; n < (NB1 ? NB1 : 1); ++n) acct[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA is invisible to the compiler's wait counts
        cp_barrier();

        auto gemm = [&](int g) {
            const u32x4 *slab = cp_smem + (g % 3) * CP_SLAB;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int WD = (CN > 0 && NW == 16) ? 1 : CP_WD;   // A fragments through a ring of WD k-steps (the 16-wave pair form has 128 registers)
            u32x4 wf[WD][4];
            auto read_w = [&](int ks, u32x4 (&w)[4]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = slab[(i * 16 + l15) * CP_CPR + ((ks * 4 + q) ^ l15)];
            };
#pragma unroll
            for (int d = 0; d < WD - 1; ++d) read_w(d, wf[d]);
#pragma unroll
            for (int ks = 0; ks < CP_KS; ++ks) {
                if (ks + WD - 1 < CP_KS) read_w(ks + WD - 1, wf[(ks + WD - 1) % WD]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = cp_mfma<F16>(wf[ks % WD][i], xf[ks], acc[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // acc[i][r] = channel g * 64 + q * 8 + (i >> 1) * 32 + (i & 1) * 4 + r of this lane's pixel
        auto epilogue = [&](int g) {
            const int c0 = g * 64 + q * 8;
            float v[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 bias = *(const f32x4 *)(sBias + c0 + (i >> 1) * 32 + (i & 1) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = i * 4 + r;
                    v[e] = acc[i][r] + bias[r];
                    const uint32_t rw = rr[e / 8][(e % 8) / 2];
                    v[e] += cp_unpack<F16>((e & 1) ? (rw >> 16) : (rw & 0xffffu));
                    v[e] = fmaxf(v[e], 0.0f);
                }
            }
            u32x4 yf[2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) yf[k][e] = usot_pack2_lp<F16>(v[k * 8 + 2 * e], v[k * 8 + 2 * e + 1]);
            if (full || pm < (long)p.M) {
                *(u32x4 *)(p.y + pm * CP_CO + c0) = yf[0];
                *(u32x4 *)(p.y + pm * CP_CO + c0 + 32) = yf[1];
            }
            // ---- second GEMM, this group's 64 k: the lane's rounded outputs are its B fragments (k-step s = run s)
            if constexpr (NB1 > 0) {
                const u32x4 *slab1 = cp_smem + (g % 3) * CP_SLAB + Cf::SLAB0;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int n = 0; n < NB1; ++n) {
                        const u32x4 w1f = slab1[(n * 16 + l15) * 8 + ((s2 * 4 + q) ^ ((l15 >> 1) & 7))];
                        acct[n] = cp_mfma<F16>(w1f, yf[s2], acct[n]);
                    }
            }
        };
        // one interval.  Leaders: GEMM(k), epilogue(k), residual of k + 1; trailers: epilogue(k - 1), residual of k, GEMM(k).
        auto interval = [&](int k) {
            if (k + 1 < CP_G) issue_slab(k + 1, (k + 1) % 3);
            bool stored = true;
            if constexpr (!trail) {
                gemm(k);
                epilogue(k);
                load_res(k + 1);
            } else {
                if (k > 0) { epilogue(k - 1); load_res(k); }
                else stored = false;
                gemm(k);
            }
            // slab k + 1 (issued at the top of this interval) must have landed before the barrier: at most the vector-memory
            // operations issued AFTER it may be outstanding - this interval's stores (2) and residual loads (2)
            if (!full) cp_wait_vm(0);
            else       cp_wait_vm(stored ? 4 : 0);
            cp_barrier();
        };
#pragma unroll 1
        for (int k = 0; k < CP_G; ++k) interval(k);
        if constexpr (trail) epilogue(CP_G - 1);
        // ---- epilogue of the second GEMM: acct[n][r] = T channel (n >> 2) * 64 + ((n >> 1) & 1) * 32 + q * 8 + (n & 1) * 4 + r
        if constexpr (NB1 > 0) {
#pragma unroll
            for (int nb = 0; nb < NB1 / 4; ++nb) {
                const int t0 = nb * 64 + q * 8;
                float v[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 b1v = *(const f32x4 *)(sBias + CP_CO + t0 + (i >> 1) * 32 + (i & 1) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[i * 4 + r] = acct[nb * 4 + i][r] + b1v[r];
                        if (p.act2 == USOT_ACT_RELU) v[i * 4 + r] = fmaxf(v[i * 4 + r], 0.0f);
                    }
                }
                if (pm < (long)p.M) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        u32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = usot_pack2_lp<F16>(v[k * 8 + 2 * e], v[k * 8 + 2 * e + 1]);
                        *(u32x4 *)(p.t + pm * CN + t0 + k * 32) = o;
                    }
                }
            }
        }
    };
    // a workgroup's waves are dealt round-robin to the four SIMDs: every SIMD hosts as many leaders as trailers
    if (wave >= NW / 2) run(std::true_type{});
    else                   run(std::false_type{});

    // ------------------------------------------------------------------ phase 5 (CM = 256, CN = 256): the NEXT block's conv1 on this panel
    // T[BM][CN] = act2(Y . w1^T + b1), K = 4 CM.  The panel's Y rows were stored by this workgroup's own waves a moment ago: after
    // vmcnt(0) + a barrier they are in L2 (write-through L1; no line of Y was ever loaded by this CU), the k-loop streams them back
    // by LDS-DMA - the standalone conv1 launch (a cold single-round kernel: 42 us for 157 MB) and its read of Y from HBM disappear.
    // Same k order as the tiled kernels: bit-identical to usot_conv2d_lp on Y.
    if constexpr (Cf::P5) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        constexpr int TN5 = Cf::TN5, WI5 = Cf::WI5;
        f32x4 acc5[TN5][TM];
#pragma unroll
        for (int i = 0; i < TN5; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc5[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int lr = tid >> 3;
        const int kc = (tid & 7) ^ cp_swz(lr);
        const uint16_t *yp[XI];
        bool yok[XI];
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const long m = (long)bm0 + lr + RPP * i;
            yok[i] = m < (long)p.M;
            yp[i] = p.y + (yok[i] ? m : 0) * CP_CO + kc * 8;
        }
        const uint16_t *w1p[WI5];
#pragma unroll
        for (int i = 0; i < WI5; ++i) w1p[i] = p.w1 + (long)(lr + RPP * i) * CP_CO + kc * 8;
        // THREE activation stages, two filter stages (all of the LDS, as tile 36 of csrc/conv_bf16.hip): Y comes back from the
        // Infinity Cache / HBM (an XCD's 32 panels = 16 MB have long left its 4 MB L2), so its DMA runs two k-tiles ahead with a
        // counted vmcnt; the filters are L2-resident and run one ahead.  (Two stages + vmcnt(0) per k-tile: 2.5 us per k-tile.)
        const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(wave * 8 * LDC * 16));
        auto issue_x5 = [&](int t, int slot) {
            const uint32_t bx = ldsw + (uint32_t)(slot * BM * LDC * 16);
#pragma unroll
            for (int i = 0; i < XI; ++i) dma16u(yok[i] ? yp[i] + t * 64 : p.zero, bx + (uint32_t)(RPP * i * LDC * 16));
        };
        auto issue_w5 = [&](int t, int buf) {
            const uint32_t bw = ldsw + (uint32_t)((3 * BM + buf * CN) * LDC * 16);
#pragma unroll
            for (int i = 0; i < WI5; ++i) dma16u(w1p[i] + t * 64, bw + (uint32_t)(RPP * i * LDC * 16));
        };
        const u32x4 *sX = cp_smem, *sW = cp_smem + 3 * BM * LDC;
        const int sq = cp_swz(l15);
        constexpr int NT5 = CP_CO / 64;
        static_assert(NT5 >= 3, "three-stage loop");
        issue_x5(0, 0);
        issue_w5(0, 0);
        issue_x5(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(XI) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int xs = 0;
        for (int t = 0; t < NT5; ++t) {
            const int xs2 = xs == 0 ? 2 : xs - 1;           // (t + 2) % 3
            // issue order: W(t + 1), then X(t + 2) - the wait below leaves only X(t + 2) outstanding (in-order retirement)
            if (t + 1 < NT5) issue_w5(t + 1, (t & 1) ^ 1);
            if (t + 2 < NT5) issue_x5(t + 2, xs2);
            const u32x4 *cX = sX + (xs * BM + wm * TM * 16 + l15) * LDC;
            const u32x4 *cW = sW + ((t & 1) * CN + wn * TN5 * 16 + l15) * LDC;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 wf[TN5], xq[TM];
#pragma unroll
                for (int i = 0; i < TN5; ++i) wf[i] = cW[i * 16 * LDC + ((ks * 4 + quad) ^ sq)];
#pragma unroll
                for (int j = 0; j < TM; ++j) xq[j] = cX[j * 16 * LDC + ((ks * 4 + quad) ^ sq)];
#pragma unroll
                for (int i = 0; i < TN5; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc5[i][j] = cp_mfma<F16>(wf[i], xq[j], acc5[i][j]);
            }
            if (t + 2 < NT5) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(XI) : "memory");
            else             asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            xs = xs == 2 ? 0 : xs + 1;
        }
        // T tile -> LDS in the storage type (rows of CN / 8 chunks, chunk c of row r at c ^ (r & 15)), then out in whole rows
        char *sT = (char *)cp_smem;
        constexpr int TCPR = CN / 8;
#pragma unroll
        for (int i = 0; i < Cf::TN5; ++i) {
            const int ch = wn * Cf::TN5 * 16 + i * 16 + quad * 4;
            const f32x4 b = *(const f32x4 *)(p.b1 + ch);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int row = wm * TM * 16 + j * 16 + l15;
                f32x4 v = acc5[i][j] + b;
                if (p.act2 == USOT_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                }
                u32x2 o;
                o[0] = usot_pack2_lp<F16>(v[0], v[1]);
                o[1] = usot_pack2_lp<F16>(v[2], v[3]);
                *(u32x2 *)(sT + ((row * TCPR + ((ch >> 3) ^ (row & 15))) * 16 + ((ch >> 2) & 1) * 8)) = o;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < BM * TCPR / NTHR; ++it) {
            const int idx = it * NTHR + tid;
            const int row = idx / TCPR, pc = idx % TCPR;
            const long m = (long)bm0 + row;
            if (m < (long)p.M) *(u32x4 *)(p.t + m * CN + (pc ^ (row & 15)) * 8) = cp_smem[idx];
        }
    }
}

template <int NW, int CM, int CN, bool RS>
int cp_launch_rs(hipStream_t s, const CpK &p, int dtype)
{
    using Cf = CpCfg<NW, CM, CN>;
    const int usot_dv = usot_device_slot();
    if (usot_dv < 0) return USOT_ESTATE;
    static bool raised_d[USOT_MAX_DEV][2] = {};
    bool (&raised)[2] = raised_d[usot_dv];
    const void *fn = dtype ? (const void *)conv_pw_kernel<true, NW, CM, CN, RS> : (const void *)conv_pw_kernel<false, NW, CM, CN, RS>;
    if (!raised[dtype]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS) != hipSuccess) return USOT_ELAUNCH;
        raised[dtype] = true;
    }
    if (dtype) hipLaunchKernelGGL((conv_pw_kernel<true, NW, CM, CN, RS>), dim3(p.npanels), dim3(Cf::NTHR), Cf::LDS, s, p);
    else       hipLaunchKernelGGL((conv_pw_kernel<false, NW, CM, CN, RS>), dim3(p.npanels), dim3(Cf::NTHR), Cf::LDS, s, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

template <int NW, int CM, int CN>
int cp_launch(hipStream_t s, const CpK &p, int dtype)
{
    return p.rs ? cp_launch_rs<NW, CM, CN, true>(s, p, dtype) : cp_launch_rs<NW, CM, CN, false>(s, p, dtype);
}

// conv2's descriptor -> kernel arguments (everything but the 1x1 banks); returns the panel size, 0 = invalid
int cp_fill(const usot_conv_desc *c2, int CM, CpK &p)
{
    if (!c2 || !c2->x || !c2->w || !c2->bias) return 0;
    if (c2->Cin != CM || c2->Cout != CM || c2->act != USOT_ACT_RELU || c2->N <= 0) return 0;
    if (c2->groups > 1 || c2->ksplit > 1 || c2->res || c2->KH <= 0 || c2->KW <= 0 || c2->stride <= 0) return 0;
    const int oh = (c2->H + 2 * c2->pad_h - c2->dil_h * (c2->KH - 1) - 1) / c2->stride + 1;
    const int ow = (c2->W + 2 * c2->pad_w - c2->dil_w * (c2->KW - 1) - 1) / c2->stride + 1;
    if (oh != c2->OH || ow != c2->OW || oh <= 0 || ow <= 0) return 0;
    if (((uintptr_t)c2->x % 16) || ((uintptr_t)c2->w % 16) || ((uintptr_t)c2->bias % 16)) return 0;
    const long M = (long)c2->N * oh * ow;
    if (M > 0x7fffffffL - 256) return 0;
    const int usot_dv = usot_device_slot();
    if (usot_dv < 0) return 0;
    static const uint16_t *zero_page_d[USOT_MAX_DEV] = {};
    const uint16_t *&zero_page = zero_page_d[usot_dv];
    if (!zero_page) {
        void *zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(cp_zero16)) != hipSuccess || !zp) return 0;
        zero_page = (const uint16_t *)zp;
    }
    // fewer than 192 panels of 256 pixels (batch 32 at layer2 / layer3 resolution: 121) would leave half the chip idle: panels of
    // 128 pixels then (8 wavefronts; c2->tile & 3 = 1 / 2 forces the 256- / 128-pixel form: tests)
    const int force = c2->tile & 3;
    const bool small = force == 2 || (force != 1 && usot_conv_pw_pixels(M) == 128);
    // the row-shared k-loop of phase 1 where the geometry allows it (c2->tile & 4: the per-tap loop anyway: tests, A/B)
    p.rs = !(c2->tile & 4) && c2->KH == 3 && c2->KW == 3 && c2->stride == 1 && c2->pad_h == c2->dil_h && c2->pad_w == c2->dil_w &&
           c2->dil_w >= 1 && c2->dil_w <= 4 && ow == c2->W && oh == c2->H &&
           (long)c2->N * c2->H * c2->W * CM < 0x7fffffffL;      // 32-bit element offsets in that loop
    const int bm = small ? 128 : 256;
    p.x = (const uint16_t *)c2->x; p.w2 = (const uint16_t *)c2->w; p.zero = zero_page; p.b2 = c2->bias;
    p.H = c2->H; p.W = c2->W; p.OW = ow; p.KW = c2->KW; p.stride = c2->stride; p.pad_h = c2->pad_h; p.pad_w = c2->pad_w;
    p.dil_h = c2->dil_h; p.dil_w = c2->dil_w;
    p.M = (int)M; p.P = oh * ow; p.KT = c2->KH * c2->KW * (CM / 64); p.npanels = (int)((M + bm - 1) / bm);
    return bm;
}

}  // namespace

extern "C" int usot_conv_pw_supported(int Cin, int CM, int CO)
{
    return Cin == CM && CO == 4 * CM && (CM == 256 || CM == 128);
}

extern "C" int usot_conv_pw_pair_supported(int CM, int CO, int CN)
{
    return (CM == 128 && CO == 512 && (CN == 128 || CN == 256)) || (CM == 256 && CO == 1024 && CN == 256);
}

/* pixels per panel (= per workgroup) the launcher would use for M pixels.  A launch is one workgroup per CU: 256-pixel panels when
 * they fill most of a round (192 ... CUs panels) or many rounds; 128-pixel panels when there are too few (batch 32) or when the
 * 256-pixel panels would spill into a mostly empty second round (CUs < panels < 1.5 CUs: e.g. 273 at batch 64 of 271 x 271 crops). */
extern "C" int usot_conv_pw_pixels(int64_t M)
{
    const int usot_dv = usot_device_slot() < 0 ? 0 : usot_device_slot();      // (a host-side shape query: no device is not an error here)
    static int cus_d[USOT_MAX_DEV] = {};
    int &cus = cus_d[usot_dv];
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    const int64_t n256 = (M + 255) / 256;
    return (n256 < 192 || (n256 > cus && 2 * n256 < 3 * cus)) ? 128 : 256;
}

/* Y = relu(relu(conv(x; c2->w) + c2->bias) . w3^T + b3 + res): see usot_hip.h */
extern "C" int usot_conv_pw_lp(void *stream, const usot_conv_desc *c2, const void *w3, const float *b3, const void *res, void *y,
                               int dtype)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    if (!c2 || !w3 || !b3 || !res || !y || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    if (!usot_conv_pw_supported(c2->Cin, c2->Cout, 4 * c2->Cout)) return USOT_EINVAL;
    const void *ptrs[] = {w3, b3, res, y};
    for (const void *q : ptrs)
        if ((uintptr_t)q % 16) return USOT_EINVAL;
    CpK p = {};
    const int bm = cp_fill(c2, c2->Cout, p);
    if (!bm) return USOT_EINVAL;
    p.w3 = (const uint16_t *)w3; p.res = (const uint16_t *)res; p.b3 = b3; p.y = (uint16_t *)y;
    hipStream_t s = (hipStream_t)stream;
    if (c2->Cout == 256) return bm == 128 ? cp_launch<8, 256, 0>(s, p, dtype) : cp_launch<16, 256, 0>(s, p, dtype);
    return bm == 128 ? cp_launch<8, 128, 0>(s, p, dtype) : cp_launch<16, 128, 0>(s, p, dtype);
}

/* the pair form: ... and T = act2(Y . w1^T + b1), the NEXT block's conv1, in the same launch.  d: w3p = w3 [CO][CM], b3, res, y,
 * w1 [CN][CO], b1, t, M (= conv2's output pixels), CM, CO, CN, act2; d->t2 is ignored */
extern "C" int usot_conv_pw_pair_lp(void *stream, const usot_conv_desc *c2, const usot_pw_pair_desc *d, int dtype)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    if (!c2 || !d || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    if (!d->w3p || !d->b3 || !d->res || !d->y || !d->w1 || !d->b1 || !d->t) return USOT_EINVAL;
    if (d->act2 != USOT_ACT_NONE && d->act2 != USOT_ACT_RELU) return USOT_EINVAL;
    if (!usot_conv_pw_pair_supported(d->CM, d->CO, d->CN) || c2->Cout != d->CM) return USOT_EINVAL;
    const void *ptrs[] = {d->w3p, d->b3, d->res, d->y, d->w1, d->b1, d->t};
    for (const void *q : ptrs)
        if ((uintptr_t)q % 16) return USOT_EINVAL;
    CpK p = {};
    const int bm = cp_fill(c2, d->CM, p);
    if (!bm || p.M != d->M) return USOT_EINVAL;
    p.w3 = (const uint16_t *)d->w3p; p.res = (const uint16_t *)d->res; p.b3 = d->b3; p.y = (uint16_t *)d->y;
    p.w1 = (const uint16_t *)d->w1; p.b1 = d->b1; p.t = (uint16_t *)d->t; p.act2 = d->act2;
    hipStream_t s = (hipStream_t)stream;
    if (d->CM == 256) return bm == 128 ? cp_launch<8, 256, 256>(s, p, dtype) : cp_launch<16, 256, 256>(s, p, dtype);
    if (d->CN == 256) return bm == 128 ? cp_launch<8, 128, 256>(s, p, dtype) : cp_launch<16, 128, 256>(s, p, dtype);
    return bm == 128 ? cp_launch<8, 128, 128>(s, p, dtype) : cp_launch<16, 128, 128>(s, p, dtype);
}
