// bf16 implicit-GEMM convolution on v_mfma_f32_16x16x32_bf16 (gfx950's 2x-K bf16 MFMA),
// fp32 accumulate, bf16 NHWC in/out.  Used for the batched backbone (BASELINE config 3:
// 64 crops, the MFMA-roofline run); the batch-1 tracking path stays fp32 (1e-4 parity).
//
// Same GEMM orientation as the fp32 kernel (weights = MFMA A operand, activations = B):
// a lane's four accumulator registers are four consecutive output channels of one pixel,
// so bias / residual / ReLU / the bf16 pack are per-lane and the store is 8 bytes.
// A lane feeds one MFMA with 8 consecutive k (16 bytes) of its row: one ds_read_b128 per
// operand per 16x16x32 step, quad q of the wave owning k-chunk q.  k-tile = 64 bf16 (128 B
// per row, XOR-swizzled in LDS), register-staged double buffering as in conv_igemm_f32.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
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
    int act, act2, act_split;      // channels < act_split get act, the rest act2
    int groups;                    // independent problems of identical geometry (the three towers)
    long x_gs, w_gs, y_gs;         // element strides between groups (y_gs in OUTPUT elements)
    int b_gs;
    int out_f32;       // store the result as fp32 (the neck output that feeds the fp32 heads)
    int M, K, KT, cchunks, MT, NT, P;
    int nfast;         // tile order: 1 = the N-tiles of one pixel tile are adjacent (same XCD, back to back)
    const uint16_t *zero;   // 16 zero bytes: the source of padding taps for the LDS-DMA paths (a kernel argument, so that its
                            // address lives in SGPRs: as a __device__ symbol its GOT load was re-issued inside the k-loop)
};

__device__ __forceinline__ int xcd_remap_b(int b, int total)
{
    const int q = total >> 3, r = total & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ uint32_t f2bf(float f)          // round to nearest even (v_cvt_pk_bf16_f32)
{
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ float bf2f(uint32_t h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ uint32_t f2h(float f) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float h2f(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
template <bool F16> __device__ __forceinline__ uint32_t pack_lp(float f) { return F16 ? f2h(f) : f2bf(f); }
template <bool F16> __device__ __forceinline__ float unpack_lp(uint32_t h) { return F16 ? h2f(h) : bf2f(h); }

__device__ __forceinline__ float act_lp(float v, int a)
{
    switch (a) {
    case USOT_ACT_RELU: return fmaxf(v, 0.0f);
    case USOT_ACT_EXP:  return expf(v);
    case USOT_ACT_CONF: return expf(fminf(fmaxf(v, 0.0f), 4.0f));
    default:            return v;
    }
}

constexpr int BKB = 64;            // k-tile in bf16 elements
// LDS rows are 8 x 16-byte chunks (64 elements), unpadded, with the chunk index XOR-swizzled by
// f(row) = (row >> 1) & 7.  ds_read_b128 is served in the lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS table), 64 banks x 4 B: with a lane reading
// chunk (4*ks + quad) of row l15 this swizzle gives each group 16 distinct 16-byte slots of the
// 256-byte bank line (the former 144-byte padded pitch was 2-way conflicted in 7 of 8 slots).
// The staging writes (8 consecutive lanes = one row's 8 chunks) stay conflict-free.
constexpr int LDC = 8;
__device__ __attribute__((aligned(16))) uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};     // source of padding taps (LDS-DMA)
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// D = k-tiles of global loads in flight per lane.  D = 1: tile t+1 is fetched during the MFMAs of
// tile t and stored at the end of the iteration (the load has half an iteration to land).  D = 2:
// a second register buffer, tile t+2 is fetched while t+1 is still in flight; every load is
// unconditional (padding taps read a valid address and are zeroed at the LDS store; past the last
// tile the last one is fetched again) so the compiler's vmcnt count stays exact and the store of
// tile t+1 waits with vmcnt(XI + WI), not 0.
// ST = LDS stages.  ST = 3 (with D = 0 only): the DMA runs two k-tiles ahead, one raw s_barrier per
// k-tile and counted s_waitcnt vmcnt(N) instead of __syncthreads (whose workgroup release drains
// every DMA in flight); used by the 8-wavefront 256x128 tiles, which also need 25 % fewer L1/LDS
// bytes per MFMA than 128x128 (a 128x128x64 step moves 32 KB per 512 MFMA cycles = 64 B/clk/CU,
// the whole vector-L1 rate).
// MF = MFMA tile edge: 16 (v_mfma_f32_16x16x32) or 32 (v_mfma_f32_32x32x16: twice the FLOPs per instruction for the
// same two 16-byte fragments, i.e. half the ds_read_b128 traffic per FLOP, and a higher issue ceiling — 2.38 vs
// 2.08 PFLOP/s in the MFMA microbenchmarks of MI355X_MICROARCH.md)
// PF (D = 0, ST = 2, MF = 16 only; round 4): bit 0 = the two-stage loop with inline-asm DMA, one raw s_barrier per k-tile and a
// COUNTED s_waitcnt vmcnt (so that the prefetches below may stay in flight across the barrier); bit 1 = pull the filter lines of
// k-tile t + 3 into this XCD's L2 while tile t is multiplied (a 4-byte LDS-DMA per 128-byte line into a dummy LDS page: no
// register destination, nothing to wait for); bit 2 = the same for the activation lines; bit 3 = the fragment reads of the
// k-tile's first half are issued BEFORE the DMA instructions of the next tile (their LDS latency then overlaps the DMA issue).
template <int BM, int BN, int WM, int WN, bool F16, int D = 1, int ST = 2, int MF = 16, int PF = 0>   // D = 0: LDS-DMA staging (below)
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 4 ? 2 : 1) void conv_igemm_bf16(const ConvB pin)
{
    ConvB p = pin;
    static_assert(PF == 0 || (D == 0 && ST == 2 && MF == 16), "PF variants are two-stage LDS-DMA kernels on the 16x16x32 MFMA");
    static_assert(WM * WN == 4 || WM * WN == 8 || WM * WN == 16, "4, 8 or 16 wavefronts");
    static_assert(ST == 2 || (ST == 3 && D == 0), "3 stages need the LDS-DMA path");
    static_assert(MF == 16 || (MF == 32 && (BM / WM) % 32 == 0 && (BN / WN) % 32 == 0), "32x32 MFMA tiles");
    constexpr int NTHR = WM * WN * 64;
    constexpr int RPP = NTHR / 8;             // tile rows one staging pass covers
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int XI = (BM + RPP - 1) / RPP, WI = (BN + RPP - 1) / RPP;
    extern __shared__ __attribute__((aligned(16))) u32x4 smem4[];
    u32x4 *sX = smem4;                      // [ST][BM][LDC]
    u32x4 *sW = smem4 + ST * BM * LDC;      // [ST][BN][LDC]

    const int tid = threadIdx.x;
    const int tiles = p.MT * p.NT;
    const int bb = xcd_remap_b(blockIdx.x, tiles * p.groups);
    const int grp = bb / tiles, b = bb - grp * tiles;
    if (p.groups > 1) {
        p.x += (long)grp * p.x_gs;
        p.w += (long)grp * p.w_gs;
        if (p.bias) p.bias += (long)grp * p.b_gs;
        p.y = (uint16_t *)((char *)p.y + (long)grp * p.y_gs * (p.out_f32 ? 4 : 2));
    }
    // The activation panel of a pixel tile is the big operand (M x K, from HBM); the filter bank is small
    // and L2-resident everywhere.  With the pixel tile as the slow index the NT tiles that share an
    // activation panel run back to back on ONE XCD (one HBM read + NT-1 L2 hits) instead of on NT
    // different XCDs (NT reads through the fabric).
    const int bn0 = p.nfast ? (b % p.NT) * BN : (b / p.MT) * BN;
    const int bm0 = p.nfast ? (b / p.NT) * BM : (b % p.MT) * BM;

    const int lr = tid >> 3;
    // D = 0 (LDS-DMA): the LDS position of a lane is fixed (wave base + lane * 16), so the swizzle
    // moves to the SOURCE: the lane at physical chunk (tid & 7) fetches logical chunk kc of its row
    const int kc = D == 0 ? ((tid & 7) ^ swz(lr)) : (tid & 7);
    int x_ih0[XI], x_iw0[XI];
    long x_nb[XI];
    bool x_ok[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int m = bm0 + lr + RPP * i;
        x_ok[i] = (lr + RPP * i < BM) && (m < p.M);
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
        const int co = bn0 + lr + RPP * i;
        wp[i] = p.w + (long)((lr + RPP * i < BN && co < p.Cout) ? co : 0) * p.K + kc * 8;
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
    u32x4 xr[D ? D : 1][XI], wr[D ? D : 1][WI];
    bool xz[D ? D : 1][XI];
    auto load_tile = [&](auto dc, bool advance) {
        constexpr int d = decltype(dc)::value;
        const int c0 = cur_cc * BKB;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            if constexpr (D == 1) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (xin[i]) v = *(const u32x4 *)(xp[i] + c0);
                xr[d][i] = v;
                xz[d][i] = true;
            } else {
                xr[d][i] = *(const u32x4 *)(xin[i] ? xp[i] + c0 : p.x + kc * 8);
                xz[d][i] = xin[i];
            }
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            wr[d][i] = *(const u32x4 *)wp[i];
            wp[i] += advance ? BKB : 0;
        }
        if (advance && ++cur_cc == p.cchunks) {
            cur_cc = 0;
            set_tap(++cur_tap);
        }
    };
    auto store_tile = [&](auto dc, int buf) {
        constexpr int d = decltype(dc)::value;
#pragma unroll
        for (int i = 0; i < XI; ++i)
            if (BM % RPP == 0 || lr + RPP * i < BM)
                sX[(buf * BM + lr + RPP * i) * LDC + (kc ^ swz(lr + RPP * i))] = (D == 1 || xz[d][i]) ? xr[d][i] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < WI; ++i)
            if (BN % RPP == 0 || lr + RPP * i < BN) sW[(buf * BN + lr + RPP * i) * LDC + (kc ^ swz(lr + RPP * i))] = wr[d][i];
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l15 = lane & 15, quad = lane >> 4;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 32x32 MFMA form: the same TN x TM x 4 registers seen as (TN/2) x (TM/2) tiles of 16.  Tile (i, j), register
    // group rg (4 consecutive registers) of lane (l31 = lane & 31, h = lane >> 5) = channels i*32 + rg*8 + h*4 .. +3
    // of pixel j*32 + l31 (cdna_hip_programming.md section 3: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
    constexpr int TN2 = MF == 32 ? TN / 2 : 1, TM2 = MF == 32 ? TM / 2 : 1;
    f32x16 acc32[TN2][TM2];
    if constexpr (MF == 32) {
#pragma unroll
        for (int i = 0; i < TN2; ++i)
#pragma unroll
            for (int j = 0; j < TM2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
    }
    const int l31 = lane & 31, hh = lane >> 5;
    // every 4-register piece of this wave's accumulators: f(pixel row in the wave's block, channel in the wave's block, value)
    auto for_each_piece = [&](auto f) {
        if constexpr (MF == 32) {
#pragma unroll
            for (int j = 0; j < TM2; ++j)
#pragma unroll
                for (int i = 0; i < TN2; ++i)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        f(j * 32 + l31, i * 32 + rg * 8 + hh * 4,
                          f32x4{acc32[i][j][rg * 4], acc32[i][j][rg * 4 + 1], acc32[i][j][rg * 4 + 2], acc32[i][j][rg * 4 + 3]});
        } else {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i) f(j * 16 + l15, i * 16 + quad * 4, acc[i][j]);
        }
    };

    auto mma_tile = [&](int cur) {
#ifdef USOT_LPABL_NOMMA      // scripts/ablate_lp.py: timing builds with parts of the kernel removed
        return;
#endif
        if constexpr (MF == 32) {
            const u32x4 *cX = sX + (cur * BM + wm * TM * 16 + l31) * LDC;
            const u32x4 *cW = sW + (cur * BN + wn * TN * 16 + l31) * LDC;
            const int sq = swz(l31);                   // rows differ from l31 by multiples of 32
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {           // four 16-deep slices of the 64-deep k-tile
                u32x4 wf[TN2], xf[TM2];
#pragma unroll
                for (int i = 0; i < TN2; ++i) wf[i] = cW[i * 32 * LDC + ((ks * 2 + hh) ^ sq)];
#pragma unroll
                for (int j = 0; j < TM2; ++j) xf[j] = cX[j * 32 * LDC + ((ks * 2 + hh) ^ sq)];
#pragma unroll
                for (int i = 0; i < TN2; ++i)
#pragma unroll
                    for (int j = 0; j < TM2; ++j) {
                        if constexpr (F16)
                            acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[i]),
                                                                                 __builtin_bit_cast(f16x8, xf[j]), acc32[i][j], 0, 0, 0);
                        else
                            acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[i]),
                                                                                  __builtin_bit_cast(bf16x8, xf[j]), acc32[i][j], 0, 0, 0);
                    }
            }
            return;
        }
        const u32x4 *cX = sX + (cur * BM + wm * TM * 16 + l15) * LDC;
        const u32x4 *cW = sW + (cur * BN + wn * TN * 16 + l15) * LDC;
        const int sq = swz(l15);                       // rows differ from l15 by multiples of 16
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
                for (int j = 0; j < TM; ++j) {
                    if constexpr (F16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[i]),
                                                                           __builtin_bit_cast(f16x8, xf[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i]),
                                                                            __builtin_bit_cast(bf16x8, xf[j]), acc[i][j], 0, 0, 0);
                }
        }
    };

    // D = 0: global_load_lds_dwordx4 (LDS-DMA).  Register staging costs a ds_write_b128 pass that
    // runs at ~79 B/clk/CU (13 cycles per wave instruction, MI355X_MICROARCH.md): measured, the
    // stores alone were ~180 us of b7.ds's 690.  The DMA writes wave-base + lane*16, i.e. 8 rows x
    // 8 chunks per instruction in exactly this tile's row-major layout.
    auto issue_tile = [&](int buf, bool advance) {
#ifdef USOT_LPABL_NOLOAD
        return;
#endif
        const int c0 = cur_cc * BKB;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const uint16_t *src = xin[i] ? xp[i] + c0 : p.zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(sX + (buf * BM + wave * 8 + RPP * i) * LDC),
                                             16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)wp[i],
                                             (__attribute__((address_space(3))) void *)(sW + (buf * BN + wave * 8 + RPP * i) * LDC),
                                             16, 0, 0);
            wp[i] += advance ? BKB : 0;
        }
        if (advance && ++cur_cc == p.cchunks) {
            cur_cc = 0;
            set_tap(++cur_tap);
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, D >= 2 ? 1 : 0>;
    (void)sizeof(I1);
    // asm LDS-DMA: invisible to the compiler's waitcnt bookkeeping (no vmcnt(0) before the next
    // ds_read or barrier); completion is counted by hand below
    auto dma16 = [&](const uint16_t *src, const u32x4 *dst) {
#ifdef USOT_LPABL_NOLOAD
        return;
#endif
        const uint32_t lds = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)dst);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
    };
    auto issue_tile_asm = [&](int buf, bool advance) {
        const int c0 = cur_cc * BKB;
#pragma unroll
        for (int i = 0; i < XI; ++i)
            dma16(xin[i] ? xp[i] + c0 : p.zero, sX + (buf * BM + wave * 8 + RPP * i) * LDC);
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            dma16(wp[i], sW + (buf * BN + wave * 8 + RPP * i) * LDC);
            wp[i] += advance ? BKB : 0;
        }
        if (advance && ++cur_cc == p.cchunks) {
            cur_cc = 0;
            set_tap(++cur_tap);
        }
    };

    const int nt = p.KT;
    if constexpr (PF != 0) {
        // ---- round 4: two stages, asm DMA, counted vmcnt, L2 prefetch of the operand lines three k-tiles ahead ----
        static_assert(BM % RPP == 0 && BN % RPP == 0, "LDS-DMA stages whole 8-row groups");
        constexpr int NW = WM * WN;
        constexpr int PD = 3;                                  // prefetch distance in k-tiles
        constexpr int NPF = ((PF & 2) ? 1 : 0) + ((PF & 4) ? 1 : 0);
        // dummy LDS page behind the two stages: a prefetch is a 4-byte-per-lane LDS-DMA whose data nobody reads
        const uint32_t pf_lds = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem4 + (uint32_t)(2 * (BM + BN) * LDC * 16));
        auto pf4 = [&](const uint16_t *src) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(pf_lds) : "memory");
        };
        // filter lines: a wave covers BN / NW rows of the tile's bank, 64 / (BN / NW) lanes per 128-byte line
        constexpr int WRW = BN / NW, XRW = BM / NW;
        static_assert(WRW <= 64 && XRW <= 64 && 64 % WRW == 0 && 64 % XRW == 0, "prefetch lane map");
        const uint16_t *wpf = nullptr;
        if constexpr ((PF & 2) != 0) {
            int co = bn0 + wave * WRW + lane / (64 / WRW);
            if (co >= p.Cout) co = p.Cout - 1;
            wpf = p.w + (long)co * p.K + (nt > PD ? PD * BKB : 0);
        }
        // activation lines of k-tile t + PD: pixel row, tap and channel chunk tracked separately from the DMA's
        int pih0 = 0, piw0 = 0, pf_cc = 0, pf_kh = 0, pf_kw = 0;
        uint32_t pnb = 0;
        if constexpr ((PF & 4) != 0) {
            int m = bm0 + wave * XRW + lane / (64 / XRW);
            if (m >= p.M) m = p.M - 1;
            const int n = m / p.P, pix = m - n * p.P;
            const int oh = pix / p.OW, ow = pix - oh * p.OW;
            pih0 = oh * p.stride - p.pad_h;
            piw0 = ow * p.stride - p.pad_w;
            pnb = (uint32_t)n * (uint32_t)(p.H * p.W) * (uint32_t)p.Cin;       // < 2^31 elements (checked by the launcher)
            const int tap = PD / p.cchunks;
            pf_cc = PD - tap * p.cchunks;
            pf_kh = tap / p.KW;
            pf_kw = tap - pf_kh * p.KW;
        }
        auto prefetch = [&](bool more) {                       // more: a k-tile t + PD + 1 exists
            if constexpr ((PF & 2) != 0) {
                pf4(wpf);
                wpf += more ? BKB : 0;
            }
            if constexpr ((PF & 4) != 0) {
                const int ih = pih0 + pf_kh * p.dil_h, iw = piw0 + pf_kw * p.dil_w;
                const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const uint32_t off = pnb + (uint32_t)(ih * p.W + iw) * (uint32_t)p.Cin + (uint32_t)(pf_cc * BKB);
                pf4(p.x + (ok ? off : 0u));
                if (more && ++pf_cc == p.cchunks) {
                    pf_cc = 0;
                    if (++pf_kw == p.KW) { pf_kw = 0; ++pf_kh; }
                }
            }
        };
        // DMA with the LDS destination as a plain 32-bit LDS address (the generic -> LDS pointer casts of dma16 cost ~40 scalar /
        // vector instructions per k-tile in front of the four DMA instructions)
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem4;
        const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(wave * 8 * LDC * 16));
        auto dma16u = [&](const uint16_t *src, uint32_t lds) {
#ifdef USOT_LPABL_NOLOAD
            return;
#endif
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
        };
        auto issue_tile_pf = [&](int buf, bool advance) {
            const int c0 = cur_cc * BKB;
            const uint32_t bx = ldsw + (uint32_t)(buf * BM * LDC * 16), bw = ldsw + (uint32_t)((2 * BM + buf * BN) * LDC * 16);
#pragma unroll
            for (int i = 0; i < XI; ++i) dma16u(xin[i] ? xp[i] + c0 : p.zero, bx + (uint32_t)(RPP * i * LDC * 16));
#pragma unroll
            for (int i = 0; i < WI; ++i) {
                dma16u(wp[i], bw + (uint32_t)(RPP * i * LDC * 16));
                wp[i] += advance ? BKB : 0;
            }
            if (advance && ++cur_cc == p.cchunks) {
                cur_cc = 0;
                set_tap(++cur_tap);
            }
        };
        auto frag_base = [&](int cur, const u32x4 *&cX, const u32x4 *&cW) {
            cX = sX + (cur * BM + wm * TM * 16 + l15) * LDC;
            cW = sW + (cur * BN + wn * TN * 16 + l15) * LDC;
        };
        auto read_half = [&](const u32x4 *cX, const u32x4 *cW, int ks, u32x4 (&wf)[TN], u32x4 (&xf)[TM]) {
            const int sq = swz(l15);
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = cW[i * 16 * LDC + ((ks * 4 + quad) ^ sq)];
#pragma unroll
            for (int j = 0; j < TM; ++j) xf[j] = cX[j * 16 * LDC + ((ks * 4 + quad) ^ sq)];
        };
        auto mma_half = [&](const u32x4 (&wf)[TN], const u32x4 (&xf)[TM]) {
#ifdef USOT_LPABL_NOMMA
            return;
#endif
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
        };
        if constexpr ((PF & 16) != 0) {
            // ---- THREE stages of the activation tile, two of the filter tile (96 + 64 KB = all of the LDS at 256 x 256): for the
            // channel-reducing 1x1 convs the activations are the HBM stream (126 MB, read once) and the filters an L2-resident
            // 512 KB, so the depth goes where the latency is: X two k-tiles ahead (64 KB in flight per CU), W one.
            static_assert(NPF == 0 && (PF & 8) == 0, "the three-stage form has no prefetch / read-first variant");
            u32x4 *sW3 = smem4 + 3 * BM * LDC;                      // filters behind THREE activation stages
            const uint32_t ldsw3 = ldsw + (uint32_t)(3 * BM * LDC * 16);
            auto issue_x3 = [&](int slot, bool advance) {
                const int c0 = cur_cc * BKB;
                const uint32_t bx = ldsw + (uint32_t)(slot * BM * LDC * 16);
#pragma unroll
                for (int i = 0; i < XI; ++i) dma16u(xin[i] ? xp[i] + c0 : p.zero, bx + (uint32_t)(RPP * i * LDC * 16));
                if (advance && ++cur_cc == p.cchunks) {
                    cur_cc = 0;
                    set_tap(++cur_tap);
                }
            };
            auto issue_w3 = [&](int buf, bool advance) {
                const uint32_t bw = ldsw3 + (uint32_t)(buf * BN * LDC * 16);
#pragma unroll
                for (int i = 0; i < WI; ++i) {
                    dma16u(wp[i], bw + (uint32_t)(RPP * i * LDC * 16));
                    wp[i] += advance ? BKB : 0;
                }
            };
            issue_x3(0, nt > 1);
            issue_w3(0, nt > 1);
            if (nt > 1) issue_x3(1, nt > 2);
            if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(XI) : "memory");
            else        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int xs = 0;
            for (int t = 0; t < nt; ++t) {
                const int xs2 = xs == 0 ? 2 : xs - 1;               // (t + 2) % 3
                // issue order: W(t + 1), then X(t + 2) — the wait below leaves only X(t + 2) outstanding (in-order retirement)
                if (t + 1 < nt) issue_w3((t & 1) ^ 1, t + 2 < nt);
                if (t + 2 < nt) issue_x3(xs2, t + 3 < nt);
                const u32x4 *cX = sX + (xs * BM + wm * TM * 16 + l15) * LDC;
                const u32x4 *cW = sW3 + ((t & 1) * BN + wn * TN * 16 + l15) * LDC;
                u32x4 wf[TN], xf[TM];
                read_half(cX, cW, 0, wf, xf);
                mma_half(wf, xf);
                read_half(cX, cW, 1, wf, xf);
                mma_half(wf, xf);
                if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(XI) : "memory");
                else            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                xs = xs == 2 ? 0 : xs + 1;
            }
        } else {
        // prologue: tile 0 by DMA, the lines of tiles 1 and 2 (the DMAs of iterations 0 and 1) towards L2 behind it
        issue_tile_pf(0, nt > 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            const bool pf = t + PD < nt;
            const u32x4 *cX, *cW;
            frag_base(cur, cX, cW);
            if constexpr ((PF & 8) != 0) {
                u32x4 wf[TN], xf[TM];
                read_half(cX, cW, 0, wf, xf);
                if (t + 1 < nt) issue_tile_pf(cur ^ 1, t + 2 < nt);
                if (NPF > 0 && pf) prefetch(t + PD + 1 < nt);
                mma_half(wf, xf);
                read_half(cX, cW, 1, wf, xf);
                mma_half(wf, xf);
            } else {
                if (t + 1 < nt) issue_tile_pf(cur ^ 1, t + 2 < nt);
                if (NPF > 0 && pf) prefetch(t + PD + 1 < nt);
                mma_tile(cur);
            }
            // the DMA pieces of tile t + 1 are older than this iteration's prefetches: in-order retirement
            if (NPF > 0 && pf) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(NPF) : "memory");
            else               asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        }
    } else if constexpr (D == 0 && ST == 3) {
        static_assert(BM % RPP == 0 && BN % RPP == 0, "LDS-DMA stages whole 8-row groups");
        constexpr int NL = XI + WI;                   // DMA instructions per wave per k-tile
        // tiles 0 and 1 in flight; tile t+2 is issued into the stage tile t-1 left (every wave passed
        // barrier t-1 after its last read of it); before barrier t each wave waits until only its
        // tile t+2 pieces are outstanding, so after the barrier tile t+1 is complete for everyone
        issue_tile_asm(0, nt > 1);
        if (nt > 1) issue_tile_asm(1, nt > 2);
        if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NL) : "memory");
        else        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int st = 0;
        for (int t = 0; t < nt; ++t) {
            const int st2 = st == 0 ? 2 : st - 1;     // (t + 2) % 3
            if (t + 2 < nt) issue_tile_asm(st2, t + 3 < nt);
            mma_tile(st);
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(NL) : "memory");
            else            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            st = st == 2 ? 0 : st + 1;
        }
    } else if constexpr (D == 0) {
        static_assert(BM % RPP == 0 && BN % RPP == 0, "LDS-DMA stages whole 8-row groups");
        issue_tile(0, nt > 1);
        __syncthreads();                              // carries the vmcnt(0) that lands the DMA
        for (int t = 0; t < nt; ++t) {
            if (t + 1 < nt) issue_tile((t & 1) ^ 1, t + 2 < nt);
            mma_tile(t & 1);
            __syncthreads();
        }
    } else if constexpr (D == 1) {
        load_tile(I0{}, nt > 1);
        store_tile(I0{}, 0);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            const bool more = t + 1 < nt;
            if (more) load_tile(I0{}, t + 2 < nt);
            mma_tile(cur);
            if (more) store_tile(I0{}, cur ^ 1);
            __syncthreads();
        }
    } else {
        int lt = 0;                                   // next tile to fetch (tile j lives in buffer j & 1)
        auto load_next = [&](auto dc) { load_tile(dc, lt + 1 < nt); ++lt; };
        load_next(I0{});
        load_next(I1{});
        store_tile(I0{}, 0);
        __syncthreads();
        auto step = [&](auto dc_free, auto dc_next, int t) {
            load_next(dc_free);                       // tile t+2 into the buffer tile t left
            mma_tile(t & 1);
            if (t + 1 < nt) store_tile(dc_next, (t & 1) ^ 1);     // tile t+1: fetched one iteration ago
            __syncthreads();
        };
        int t = 0;
        for (; t + 2 <= nt; t += 2) {
            step(I0{}, I1{}, t);
            step(I1{}, I0{}, t + 1);
        }
        if (t < nt) step(I0{}, I1{}, t);
    }

    // Epilogue.  The accumulator layout gives a lane four channels of one pixel: stored
    // directly that is 8-byte pieces scattered over 16 pixel rows per instruction, and the
    // residual is read the same way (measured 1.1-1.4 TB/s on the 1x1 expansion convs, which
    // are pure HBM traffic).  Instead the fp32 tile is transposed through LDS (free after the
    // k-loop) so that every lane handles 8 consecutive channels: 16-byte residual loads and
    // 16-byte stores, 16 lanes per 256-byte row segment.
    // Tiles whose fp32 image exceeds the LDS (256x256) go through in EH channel slices.
    constexpr int EH = (BM * (BN + 4) * 4 > 144 * 1024) ? 2 : 1;
    constexpr int BNH = BN / EH;                     // channels per slice
    constexpr int OP = BNH + 4;                      // fp32 row pitch: 4*odd dwords -> conflict-free
    static_assert(EH == 1 || (BNH % (TN * 16) == 0), "a wave's channels fall into one slice");
    if (!p.out_f32 && (p.Cout & 7) == 0) {
        float *sO = (float *)smem4;
        constexpr int CPRW = BNH / 8;                // 16-byte output chunks per row
        constexpr int RPASS = NTHR / CPRW;           // rows per pass
        constexpr int NP = (BM + RPASS - 1) / RPASS;
        constexpr int CH = NP < 4 ? NP : 4;          // residual loads in flight per lane
        const int oc = tid % CPRW, orow = tid / CPRW;
#pragma unroll 1
        for (int eh = 0; eh < EH; ++eh) {
            if (eh > 0) __syncthreads();             // the previous slice has been read out
            if ((wn * TN * 16) / BNH == eh) {
                const int cl0 = wn * TN * 16 - eh * BNH;
                for_each_piece([&](int prow, int pch, f32x4 v) { *(f32x4 *)(sO + (wm * TM * 16 + prow) * OP + cl0 + pch) = v; });
            }
            __syncthreads();
            const int co = bn0 + eh * BNH + oc * 8;
            const bool cok = co < p.Cout;
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (p.bias && cok) { b0 = *(const f32x4 *)(p.bias + co); b1 = *(const f32x4 *)(p.bias + co + 4); }
#pragma unroll 1
            for (int q0 = 0; q0 < NP; q0 += CH) {
                u32x4 rr[CH];
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const int r = orow + (q0 + q) * RPASS, m = bm0 + r;
                    rr[q] = u32x4{0u, 0u, 0u, 0u};
#ifndef USOT_LPABL_NORES
                    if (p.res && cok && r < BM && m < p.M) rr[q] = *(const u32x4 *)(p.res + (long)m * p.Cout + co);
#endif
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
                    const int av = co < p.act_split ? p.act : p.act2;
                    if (av == USOT_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                    } else if (av != USOT_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = act_lp(v0[e], av); v1[e] = act_lp(v1[e], av); }
                    }
                    u32x4 o;
                    o[0] = usot_pack2_lp<F16>(v0[0], v0[1]);
                    o[1] = usot_pack2_lp<F16>(v0[2], v0[3]);
                    o[2] = usot_pack2_lp<F16>(v1[0], v1[1]);
                    o[3] = usot_pack2_lp<F16>(v1[2], v1[3]);
#ifdef USOT_LPABL_NOSTORE
                    if (o[0] == 0x12345678u)           // keeps the value live, never true in practice
#endif
                    *(u32x4 *)(p.y + (long)m * p.Cout + co) = o;
                }
            }
        }
        return;
    }

    for_each_piece([&](int prow, int pch, f32x4 v) {
        const int m = bm0 + wm * TM * 16 + prow;
        const int co = bn0 + wn * TN * 16 + pch;
        if (m >= p.M || co >= p.Cout) return;
        if (p.bias) v += *(const f32x4 *)(p.bias + co);
        if (p.res) {
            const u32x2 r = *(const u32x2 *)(p.res + (long)m * p.Cout + co);
            v[0] += unpack_lp<F16>(r[0] & 0xffffu); v[1] += unpack_lp<F16>(r[0] >> 16);
            v[2] += unpack_lp<F16>(r[1] & 0xffffu); v[3] += unpack_lp<F16>(r[1] >> 16);
        }
        const int av = co < p.act_split ? p.act : p.act2;
        if (av != USOT_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_lp(v[e], av);
        }
        if (p.out_f32) {
            *(f32x4 *)((float *)p.y + (long)m * p.Cout + co) = v;
        } else {
            u32x2 o;
            o[0] = usot_pack2_lp<F16>(v[0], v[1]);
            o[1] = usot_pack2_lp<F16>(v[2], v[3]);
            *(u32x2 *)(p.y + (long)m * p.Cout + co) = o;
        }
    });
}

struct TileB { int bm, bn; void (*fn)(const ConvB); void (*fn16)(const ConvB); int threads, stages, pf = 0; };
#define TB(bm, bn, wm, wn) { bm, bn, conv_igemm_bf16<bm, bn, wm, wn, false>, conv_igemm_bf16<bm, bn, wm, wn, true>, 256, 2 }
#define TB2(bm, bn, wm, wn) { bm, bn, conv_igemm_bf16<bm, bn, wm, wn, false, 2>, conv_igemm_bf16<bm, bn, wm, wn, true, 2>, 256, 2 }
#define TB0(bm, bn, wm, wn) { bm, bn, conv_igemm_bf16<bm, bn, wm, wn, false, 0>, conv_igemm_bf16<bm, bn, wm, wn, true, 0>, 256, 2 }
#define TB08(bm, bn, wm, wn) { bm, bn, conv_igemm_bf16<bm, bn, wm, wn, false, 0>, conv_igemm_bf16<bm, bn, wm, wn, true, 0>, wm * wn * 64, 2 }
#define TB32(bm, bn, wm, wn) { bm, bn, conv_igemm_bf16<bm, bn, wm, wn, false, 0, 2, 32>, conv_igemm_bf16<bm, bn, wm, wn, true, 0, 2, 32>, wm * wn * 64, 2 }
#define TBP(bm, bn, wm, wn, pf) { bm, bn, conv_igemm_bf16<bm, bn, wm, wn, false, 0, 2, 16, pf>, conv_igemm_bf16<bm, bn, wm, wn, true, 0, 2, 16, pf>, wm * wn * 64, 2, pf }
#define TB3(bm, bn, wm, wn) { bm, bn, conv_igemm_bf16<bm, bn, wm, wn, false, 0, 3>, conv_igemm_bf16<bm, bn, wm, wn, true, 0, 3>, wm * wn * 64, 3 }
// As in conv_igemm.hip: the default build compiles the ROUTED tiles (usot_amd/data/tuning_lp_gfx950.json, the launcher's heuristic
// below, and tile 21 - the plain form of tile 32's loop); the other ids are empty slots unless built with -DUSOT_EXPERIMENTS.
#ifdef USOT_EXPERIMENTS
#define XTB(...) __VA_ARGS__
#else
#define XTB(...) TileB{0, 0, nullptr, nullptr, 0, 0}
#endif
const TileB kTilesB[] = {
    TB(128, 128, 2, 2),   // 1
    XTB(TB(128, 64, 2, 2)),    // 2
    XTB(TB(64, 128, 2, 2)),    // 3
    TB(64, 64, 2, 2),     // 4
    TB(32, 64, 2, 2),     // 5
    XTB(TB2(128, 128, 2, 2)),  // 6: two k-tiles of loads in flight
    XTB(TB2(128, 64, 2, 2)),   // 7
    XTB(TB2(64, 128, 2, 2)),   // 8
    XTB(TB2(64, 64, 2, 2)),    // 9
    XTB(TB0(128, 128, 2, 2)),  // 10: LDS-DMA staging
    TB0(128, 64, 2, 2),   // 11
    TB0(64, 128, 2, 2),   // 12
    TB0(64, 64, 2, 2),    // 13
    TB0(32, 64, 2, 2),    // 14
    XTB(TB3(256, 128, 4, 2)),  // 15: 8 wavefronts, 3-stage LDS-DMA pipeline
    XTB(TB3(128, 256, 2, 4)),  // 16
    XTB(TB3(128, 128, 2, 2)),  // 17: 4 wavefronts, 3 stages (one workgroup per CU)
    XTB(TB08(256, 256, 2, 4)), // 18: 8 wavefronts x (128 pixels x 64 channels), 2-stage LDS-DMA
    XTB(TB08(256, 128, 4, 2)), // 19
    XTB(TB08(128, 256, 2, 4)), // 20
    TB08(256, 256, 4, 4), // 21: 16 wavefronts x (64 x 64)
    XTB(TB08(256, 256, 4, 2)), // 22: 8 wavefronts x (64 pixels x 128 channels)
    XTB(TB08(256, 128, 4, 4)), // 23: 16 wavefronts x (64 x 32)
    XTB(TB08(128, 256, 2, 8)), // 24: 16 wavefronts x (64 x 32)
    TB08(128, 128, 4, 4), // 25: 16 wavefronts x (32 x 32)
    TB32(256, 256, 4, 4), // 26: as 21 on v_mfma_f32_32x32x16 (2 x 2 tiles of 32 per wave)
    XTB(TB32(256, 256, 4, 2)), // 27: 8 wavefronts x (64 pixels x 128 channels), 32x32 MFMA
    XTB(TB32(256, 256, 2, 4)), // 28: 8 wavefronts x (128 x 64)
    XTB(TB32(128, 128, 4, 4)), // 29: as 25 (one 32x32 tile per wave)
    XTB(TB32(256, 128, 4, 2)), // 30: 8 wavefronts x (64 x 64)
    XTB(TB32(128, 256, 2, 4)), // 31
    TBP(256, 256, 4, 4, 1),   // 32: tile 21 with the asm two-stage loop (integer LDS addresses, zero page from the kernel arguments)
    XTB(TBP(256, 256, 4, 4, 3)),   // 33: + the filter lines of k-tile t + 3 pulled into L2 by a 4-byte-per-line LDS-DMA (slower: below)
    XTB(TBP(256, 256, 4, 4, 7)),   // 34: + the activation lines
    XTB(TBP(256, 256, 4, 4, 9)),   // 35: tile 32 with the fragment reads of the k-tile's first half before the DMA issue
    XTB(TBP(256, 256, 4, 4, 17)),  // 36: THREE activation stages + two filter stages (all 160 KB): the 1024 -> 256 reductions' HBM stream
    TBP(128, 128, 4, 4, 1),   // 37: tile 25 on the asm loop (layer2's 3x3 convs, N = 128: 27.0 -> 25.9 us isolated; 256 x 128 forms 25.4 / 28.4)
    // (Round 4, isolated on layer3's shortcut conv / its conv2 / layer2's shortcut conv, us per launch: tile 21 439 / 60.8 / 120;
    //  32: 427 / 60.4 / 119; 33: 468 / 65.7 / 134; 34: 504 / 69.0 / 140; 35: 424 / 59.5 / 120.  The prefetch variants test the
    //  hypothesis "a k-tile waits for the filter lines' Infinity-Cache latency": refuted — every EXTRA vector-memory instruction per
    //  wave and k-tile costs ~13 cycles of the k-tile (16 more: +200 cycles, 32 more: +430), whatever it fetches, i.e. the
    //  64 DMA instructions of a 256 x 256 x 64 k-tile account for ~830 of its ~2 970 cycles next to 2 048 cycles of MFMA.  In the
    //  backbone graph tile 32 is 10 us per step faster than 21 (2 341-2 350 vs 2 345-2 360), tile 35 6-18 us slower.)
    // (4 wavefronts x (128 x 128) on a 256 x 256 tile - a quarter of tile 21's LDS fragment bytes per MFMA, accumulators in
    //  AGPRs - measured 648 TFLOP/s on layer3's shortcut conv against 1 082-1 186 for tiles 21 / 22, 981 on the 32x32x16 MFMA:
    //  one wave per SIMD under hipcc's schedule, 512 registers and spills.  Not kept.)
    // (Round 3, "ping-pong": SQ_VALU_MFMA_BUSY_CYCLES is 67 % of the kernel on layer3's shortcut conv with tile 21 (rocprofv3
    //  --pmc; the clock under this load is 1.8 GHz, LDS array 35 % busy, no bank conflicts).  A schedule in which the two halves
    //  of the workgroup run half a k-step apart — one raw s_barrier per phase; a memory phase reads one k-step's fragments into
    //  registers, a compute phase issues its MFMAs back to back from registers — did NOT close that gap: 8 waves x (128 x 64)
    //  with the DMA pieces between the MFMAs 608 us against 510 (the 60-180-cycle issue stall of an LDS-DMA instruction lands on
    //  the only wave feeding the SIMD's matrix pipe), 16 waves x (64 x 64) in teams of two per SIMD 524 against 521, worse on
    //  K = 2304.  Issuing the DMAs between the two k-steps instead of before them: tiles 18 / 22 -3...-9 %, tile 21 +5 %.  Not kept.)
};
constexpr int kNumTilesB = sizeof(kTilesB) / sizeof(kTilesB[0]);

// fp32 -> bf16 (and back) elementwise, 8 elements per thread
template <bool F16>
__global__ __launch_bounds__(256) void cvt_f32_bf16_kernel(const float *__restrict__ src, uint16_t *__restrict__ dst, long n8)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f32x4 a = ((const f32x4 *)src)[2 * i], b = ((const f32x4 *)src)[2 * i + 1];
        u32x4 o;
        o[0] = usot_pack2_lp<F16>(a[0], a[1]); o[1] = usot_pack2_lp<F16>(a[2], a[3]);
        o[2] = usot_pack2_lp<F16>(b[0], b[1]); o[3] = usot_pack2_lp<F16>(b[2], b[3]);
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
        for (int e = 0; e < 4; ++e) o[e] = usot_pack2_lp<F16>(m[2 * e], m[2 * e + 1]);
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

#ifndef USOT_STEM_MINW
#define USOT_STEM_MINW 4
#endif
template <bool F16, bool O16 = F16>      // F16: the MFMA / staging type; O16: the storage type of the output (fp16 or bf16)
__global__ __launch_bounds__(256, USOT_STEM_MINW) void stem_pool_lp_kernel(
    const float *__restrict__ x, const u32x4 *__restrict__ wfrag, const float *__restrict__ bias,
    uint16_t *__restrict__ y, int H, int W, int OH, int OW, int PH, int PW, float mu0, float mu1, float mu2, int strip, int tiles_x)
{
    // bf16 keeps 8 significant bits: rounding the CROP to bf16 doubles the end-to-end error of the
    // whole backbone (measured 4e-2 -> 8e-2 of the feature scale), so the bf16 variant stages the
    // crop as hi + lo (two bf16, ~16 bits) and issues two MFMAs per fragment; fp16 (11 bits) does not.
    constexpr bool SPLIT = !F16;
    constexpr int PLANE = 3 * SP_IR * SP_ICP;
    __shared__ __attribute__((aligned(16))) uint16_t patch[(SPLIT ? 2 : 1) * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned char stile[SP_NBLK * 16 * SP_OPB];
    const int n = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, quad = lane >> 4;

    // filter fragments: 4 channel blocks x 6 k-steps, 16 bytes each, coalesced.  24 KB per wave: at one tile per workgroup
    // (8 192 workgroups at batch 64) that is 800 MB of L2 -> CU traffic per launch — more than the launch's HBM bytes ten times
    // over and ~half its time — so a workgroup walks a STRIP of `strip` tiles along x with the fragments resident.
    // A wave owns HALF the channels (2 blocks of 16: 48 fragment registers instead of 96 -> four workgroups per CU) and every
    // other pixel block; a pixel block's B fragments are read from LDS by two waves.
    const int cb0 = (wave & 1) * 2;
    u32x4 wf[2][6];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) wf[cb][ks] = wfrag[((cb0 + cb) * 6 + ks) * 64 + lane];
    __shared__ __attribute__((aligned(16))) float sbias[64];    // read back per pixel block: 16 registers less across the strip
    if (tid < 64) sbias[tid] = bias[tid];

    for (int tx = blockIdx.x * strip; tx < tiles_x && tx < (int)(blockIdx.x + 1) * strip; ++tx) {
    if (tx != (int)blockIdx.x * strip) __syncthreads();         // the previous tile's pool has read `stile`
    const int py0 = blockIdx.y * SP_P, px0 = tx * SP_Q;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;
    const int iy0 = 2 * sy0, ix0 = 2 * sx0;

    // input patch -> LDS in the storage type (out-of-image pixels and the pad column are zero;
    // they only ever feed stem pixels the pool masks out, or zero weights)
    const float *xn = x + (long)n * 3 * H * W;
    // Four pixels per task: a patch row is 10 segments of 4 (global_load_dwordx4 needs 4-byte alignment only), 69 rows: 690
    // tasks, 3 rounds of 256 threads, all loads issued before the first LDS store.  (One pixel per task — 11 rounds of index
    // arithmetic, 22 two-byte LDS stores per thread — was 64 us of the 121 us launch at batch 64, scripts/stem_lp_probe.py.)
    typedef f32x4 f32x4_a4 __attribute__((aligned(4)));
    constexpr int SEGS = SP_ICP / 4, NTASK = 3 * SP_IR * SEGS, NPL = (NTASK + 255) / 256;
    f32x4 pv[NPL];
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int t = tid + q * 256;
        const int row = t / SEGS, seg = t - row * SEGS;
        const int ci = row / SP_IR, py = row - ci * SP_IR;
        const int iy = iy0 + py, ix = ix0 + seg * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#ifndef USOT_SPABL_NOSTAGE     // scripts/stem_lp_probe.py: timing builds with parts of the kernel removed
        if (t < NTASK && (unsigned)iy < (unsigned)H) {
            const float *src = xn + ((long)ci * H + iy) * W + ix;
            const float mu = ci == 0 ? mu0 : (ci == 1 ? mu1 : mu2);
            if (ix >= 0 && ix + 3 < W) {
                v = *(const f32x4_a4 *)src - mu;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((unsigned)(ix + e) < (unsigned)W) v[e] = src[e] - mu;
            }
        }
#endif
        if (seg == SEGS - 1) v[3] = 0.f;             // the pad column only meets zero weights: keep it finite whatever the crop holds
        pv[q] = v;
    }
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int t = tid + q * 256;
        if (t < NTASK) {
            const uint32_t h0 = usot_pack2_lp<F16>(pv[q][0], pv[q][1]), h1 = usot_pack2_lp<F16>(pv[q][2], pv[q][3]);
            *(u32x2 *)(patch + t * 4) = u32x2{h0, h1};
            if constexpr (SPLIT) {
                const uint32_t l0 = usot_pack2_lp<F16>(pv[q][0] - unpack_lp<F16>(h0 & 0xffffu), pv[q][1] - unpack_lp<F16>(h0 >> 16));
                const uint32_t l1 = usot_pack2_lp<F16>(pv[q][2] - unpack_lp<F16>(h1 & 0xffffu), pv[q][3] - unpack_lp<F16>(h1 >> 16));
                *(u32x2 *)(patch + PLANE + t * 4) = u32x2{l0, l1};
            }
        }
    }
    __syncthreads();

#ifdef USOT_SPABL_NOMMA
    for (int blk = wave >> 1; blk < SP_NBLK && H < 0; blk += 2) {
#else
    for (int blk = wave >> 1; blk < SP_NBLK; blk += 2) {
#endif
        int pi = blk * 16 + l15;
        if (pi > SP_NPIX - 1) pi = SP_NPIX - 1;
        const int sy = pi / SP_C, sx = pi - sy * SP_C;
        // two accumulation chains per channel block (bf16: the hi and the lo plane of the crop; fp16: even and odd k-steps), so
        // that four independent MFMAs are in flight; summed at the end
        f32x4 acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[cb][0] = acc[cb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            int r = 4 * ks + quad;
            if (r > 20) r = 20;                          // zero-weight rows: any valid address
            const int ci = r / 7, kh = r - ci * 7;
            // (two copies of the patch one dword apart, so that every fragment is two aligned ds_read_b64, were slower: 103 us
            //  against 98 at batch 64, fp16 76 against 65)
            const uint32_t *src = (const uint32_t *)(patch + (ci * SP_IR + 2 * sy + kh) * SP_ICP + 2 * sx);
            const u32x4 xf = {src[0], src[1], src[2], src[3]};
            if constexpr (F16) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    acc[cb][ks & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[cb][ks]),
                                                                             __builtin_bit_cast(f16x8, xf), acc[cb][ks & 1], 0, 0, 0);
            } else {
                const uint32_t *srl = src + PLANE / 2;
                const u32x4 xl = {srl[0], srl[1], srl[2], srl[3]};
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    acc[cb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[cb][ks]),
                                                                         __builtin_bit_cast(bf16x8, xf), acc[cb][0], 0, 0, 0);
                    acc[cb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[cb][ks]),
                                                                         __builtin_bit_cast(bf16x8, xl), acc[cb][1], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            f32x4 v = (acc[cb][0] + acc[cb][1]) + *(const f32x4 *)(sbias + (cb0 + cb) * 16 + quad * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            u32x2 o;
            o[0] = usot_pack2_lp<O16>(v[0], v[1]);
            o[1] = usot_pack2_lp<O16>(v[2], v[3]);
            *(u32x2 *)(stile + (blk * 16 + l15) * SP_OPB + ((cb0 + cb) * 16 + quad * 4) * 2) = o;
        }
    }
    __syncthreads();

    // 3x3 / stride 2 / pad 1 max-pool: one pooled pixel x 8 channels per lane
    const int pp = tid >> 3, c8 = tid & 7;
    const int ppy = pp / SP_Q, ppx = pp - ppy * SP_Q;
    const int py = py0 + ppy, px = px0 + ppx;
    bool live = py < PH && px < PW;
#ifdef USOT_SPABL_NOPOOL
    live = live && H < 0;
#endif
    if (live) {
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
                m[2 * e] = fmaxf(m[2 * e], unpack_lp<O16>(v[e] & 0xffffu));
                m[2 * e + 1] = fmaxf(m[2 * e + 1], unpack_lp<O16>(v[e] >> 16));
            }
        }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = usot_pack2_lp<O16>(m[2 * e], m[2 * e + 1]);
    *(u32x4 *)(y + ((((long)n * PH + py) * PW + px) * 64 + c8 * 8)) = o;
    }
    }
}

}  // namespace

extern "C" int usot_conv_bf16_tile_count(void) { return kNumTilesB; }
extern "C" int usot_conv_bf16_tile_built(int tile) { return (tile >= 1 && tile <= kNumTilesB && kTilesB[tile - 1].fn) ? 1 : 0; }

/* bf16|fp16 NHWC conv: x/w/res/y in the storage type (uint16), bias fp32.  Uses the fields N..dil_w,
 * act/act2/act_split, groups (+ x_gs, w_gs, b_gs, y_gs), tile of usot_conv_desc; y dense NHWC
 * [groups][N][OH][OW][Cout]; res same layout (groups == 1 only); Cin % 64 == 0, Cout % 4 == 0;
 * ksplit / nchw / channel-offset outputs are fp32-path features. */
extern "C" int usot_conv2d_lp(void *stream, const usot_conv_desc *d, int dtype, int out_f32)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    if (dtype != 0 && dtype != 1) return USOT_EINVAL;
    if (!d || !d->x || !d->w || !d->y) return USOT_EINVAL;
    if (d->Cin <= 0 || (d->Cin % BKB) || d->Cout <= 0 || (d->Cout & 3) || d->N <= 0) return USOT_EINVAL;
    if (d->ksplit > 1 || d->y_nchw || d->groups < 0 || (d->groups > 1 && d->res)) return USOT_EINVAL;
    if (d->act < USOT_ACT_NONE || d->act > USOT_ACT_CONF) return USOT_EINVAL;
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
    p.act_split = d->act_split > 0 ? d->act_split : d->Cout; p.act2 = d->act_split > 0 ? d->act2 : d->act;
    p.groups = d->groups > 1 ? d->groups : 1;
    p.x_gs = d->x_gs; p.w_gs = d->w_gs; p.y_gs = d->y_gs; p.b_gs = (int)d->b_gs;
    p.P = d->OH * d->OW; p.M = d->N * p.P; p.K = d->KH * d->KW * d->Cin;
    p.cchunks = d->Cin / BKB; p.KT = d->KH * d->KW * p.cchunks;
    int tile = d->tile;
    if (tile == 0) {
        const long b128 = (long)((p.M + 127) / 128) * ((d->Cout + 127) / 128);
        tile = (b128 >= 512 && d->Cout >= 128) ? 1 : ((long)((p.M + 63) / 64) * ((d->Cout + 63) / 64) >= 512 ? 4 : 5);
    }
    if (tile < 1 || tile > kNumTilesB) return USOT_EINVAL;
    const TileB &tc = kTilesB[tile - 1];
    if (!tc.fn) return USOT_ENOTBUILT;
    p.MT = (p.M + tc.bm - 1) / tc.bm;
    p.NT = (d->Cout + tc.bn - 1) / tc.bn;
    p.nfast = 1;
    const long blocks = (long)p.MT * p.NT * p.groups;
    if (blocks <= 0 || blocks > 0x7fffffffL) return USOT_EINVAL;
    static const uint16_t *zero_page_d[USOT_MAX_DEV] = {};
    const uint16_t *&zero_page = zero_page_d[usot_dv];
    if (!zero_page) {
        void *zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero16)) != hipSuccess || !zp) return USOT_ELAUNCH;
        zero_page = (const uint16_t *)zp;
    }
    p.zero = zero_page;
    if (tc.pf && ((long)d->N * d->H * d->W * d->Cin >= 0x7fffffffL)) return USOT_EINVAL;     // 32-bit prefetch offsets
    size_t lds = (size_t)tc.stages * (tc.bm + tc.bn) * LDC * 16 + ((tc.pf & 6) ? 256 : 0);      // + the prefetches' dummy page
    if (tc.pf & 16) lds = (size_t)(3 * tc.bm + 2 * tc.bn) * LDC * 16;
    size_t lds_out = (size_t)tc.bm * (tc.bn + 4) * 4;            // fp32 staging tile of the epilogue
    if (lds_out > 144 * 1024) lds_out = (size_t)tc.bm * (tc.bn / 2 + 4) * 4;   // two channel slices
    if (lds_out > lds) lds = lds_out;
    if (lds > 64 * 1024) {
        static bool raised_d[USOT_MAX_DEV][2][kNumTilesB + 1] = {};
        bool (&raised)[2][kNumTilesB + 1] = raised_d[usot_dv];
        if (!raised[dtype][tile]) {
            if (hipFuncSetAttribute((const void *)(dtype ? tc.fn16 : tc.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return USOT_ELAUNCH;
            raised[dtype][tile] = true;
        }
    }
    hipLaunchKernelGGL(dtype ? tc.fn16 : tc.fn, dim3((unsigned)blocks), dim3(tc.threads), lds, (hipStream_t)stream, p);
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
    if (!x || !wfrag || !bias || !y || N <= 0 || H < 7 || W < 7 || dtype < 0 || dtype > 2) return USOT_EINVAL;
    if (OH != (H - 7) / 2 + 1 || OW != (W - 7) / 2 + 1) return USOT_EINVAL;
    if (PH != (OH + 2 - 3) / 2 + 1 || PW != (OW + 2 - 3) / 2 + 1) return USOT_EINVAL;
    if (((uintptr_t)wfrag % 16) || ((uintptr_t)y % 16) || ((uintptr_t)bias % 16) || N > 65535) return USOT_EINVAL;
    const int tiles_x = usot_cdiv(PW, SP_Q), tiles_y = usot_cdiv(PH, SP_P);
    // strip length: the longest of 4, 2, 1 that still leaves >= 6 workgroups per CU (three are resident)
    int strip = 4;                                              // (1 / 2 / 4 / 8 measured at batch 64: 106 / 103 / 98 / 99 us)
    while (strip > 1 && (long)usot_cdiv(tiles_x, strip) * tiles_y * N < 6 * 256) strip >>= 1;
    dim3 grid(usot_cdiv(tiles_x, strip), tiles_y, N);
    if (dtype == 2) hipLaunchKernelGGL((stem_pool_lp_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, x, (const u32x4 *)wfrag, bias, (uint16_t *)y, H, W, OH, OW, PH, PW, mu0, mu1, mu2, strip, tiles_x);
    else if (dtype) hipLaunchKernelGGL(stem_pool_lp_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (const u32x4 *)wfrag, bias, (uint16_t *)y, H, W, OH, OW, PH, PW, mu0, mu1, mu2, strip, tiles_x);
    else       hipLaunchKernelGGL(stem_pool_lp_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (const u32x4 *)wfrag, bias, (uint16_t *)y, H, W, OH, OW, PH, PW, mu0, mu1, mu2, strip, tiles_x);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}
