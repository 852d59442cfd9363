// Pointwise (1x1) convolutions of the batched low-precision backbone (BASELINE config 3), pixel-stationary:
//
//     Y[M][N]  = act(X[M][K] . W^T + b (+ R))       a bottleneck's conv3 + BN (+ residual) + ReLU, or layer1's 1x1
//                                                   shortcut (modules.py:48-56,108-113); K = 64 | 128 | 256, N = 4 K
//     T[M][CN] = act2(Y . W1^T + b1)                optionally, in the same launch: the NEXT block's conv1 + BN + ReLU
//                                                   (modules.py:40-42) or the neck's 1x1 (connect.py:294-300)
//
// These layers are pure HBM traffic (layer3's conv3: 283 MB per launch for 32 GFLOP) and the tiled implicit-GEMM kernel
// moved them at 3.2 TB/s: each 128 x 128 tile is a load phase, an MFMA phase and a 64 KB epilogue in sequence, X is re-read
// through L2 by eight channel tiles and W by 481 pixel tiles (500 MB of L2 -> LDS traffic per launch), and only two such
// workgroups share a CU (scripts/ablate_lp.py: load phase alone 52 us, epilogue alone 53, together 99).
//
// Here a workgroup owns a PANEL of pixels and walks the output channels in groups of 64:
//   * the panel's B operands (X fragments, 16 pixels x 32 k each) live in REGISTERS for the whole panel: a wave owns PB
//     pixel blocks, loaded once straight from global memory (K = 256, PB = 2: 64 registers) — X never touches LDS;
//   * W streams through LDS once per panel in slabs of 64 channels x K (LDS-DMA, a ring of slabs); all eight waves read
//     their A fragments from the same slab: every byte of W crosses L2 -> CU once per panel;
//   * the slab's rows are permuted — MFMA row rho of 16-channel block i is channel (i >> 1) * 32 + (rho >> 2) * 8 +
//     (i & 1) * 4 + (rho & 3) — so that a lane's sixteen accumulator registers of a group are two runs of 8 CONTIGUOUS
//     channels of one pixel, 32 channels apart: the residual is read and Y is written in 16-byte pieces, the four quads of a
//     pixel covering 64 contiguous bytes per instruction and 128 per pixel and wave, with no LDS transpose and no barrier in
//     the epilogue.  A wave's epilogue (HBM) overlaps the other waves' MFMAs; one barrier per group retires
//     a slab;
//   * (pair form) those sixteen channels, rounded to the storage type, ARE two B fragments of the second GEMM: with its k
//     axis ordered as (group, half, quad, 8) the lane that produced Y[pixel][g*64 + s*32 + q*8 .. +7] is the lane that
//     must supply it for k-step (g, s).  Y goes from the accumulators into the next MFMA without LDS, without a shuffle
//     and without ever being read back from HBM (29 % fewer bytes per pair); W1's k-slice of the group rides in the same
//     slab ring, its rows permuted like W's so that T is stored in 32-byte pieces too.
// Measured at batch 64, layer3's conv3 (M = 61 504): tiled kernel 89 us -> panel 79 (residual loaded in the epilogue) ->
// 67 (residual prefetched one group ahead, bias in LDS: a vector load in the loop would drain the prefetch, vmcnt retires
// in order); two groups ahead: no further gain.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct PanelK {
    const uint16_t *x, *w, *res, *w1;
    const float *bias, *b1;
    uint16_t *y, *t;
    int M, act, act2, npanels;
};

__device__ __forceinline__ uint32_t pn_f2bf(float f)
{
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}
template <bool F16> __device__ __forceinline__ uint32_t pn_pack(float f)
{
    return F16 ? (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f) : pn_f2bf(f);
}
template <bool F16> __device__ __forceinline__ float pn_unpack(uint32_t h)
{
    return F16 ? (float)__builtin_bit_cast(_Float16, (uint16_t)h) : __builtin_bit_cast(float, h << 16);
}
template <bool F16> __device__ __forceinline__ f32x4 pn_mfma(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else               return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// CN = 0: the single convolution.  PB = 16-pixel blocks per wave.
template <int K, int N, int CN, int PB_>
struct PanelCfg {
    static constexpr int NW = 8;                          // wavefronts
    static constexpr int PB = PB_;
    static constexpr int BM = NW * PB * 16;               // pixels per panel
    static constexpr int KS = K / 32;                     // MFMA k-steps of the first GEMM
    static constexpr int CPR = K / 8;                     // 16-byte chunks per slab row
    static constexpr int G = N / 64;                      // channel groups per panel
    static constexpr int SLAB0 = 64 * CPR;                // 16-byte chunks of W's slab (64 channels x K)
    static constexpr int SLAB1 = CN * 8;                  // ... of W1's slab (CN channels x this group's 64 k)
    static constexpr int SLAB = SLAB0 + SLAB1;
    static constexpr int NI0 = SLAB0 / (NW * 64), NI1 = SLAB1 / (NW * 64);   // DMA instructions per thread and slab
    static constexpr int NI = NI0 + NI1;
    static constexpr int S = 3;                           // slabs in the ring
    static constexpr int NB1 = CN / 16;                   // 16-channel blocks of T
    static constexpr int LDS_BYTES = S * SLAB * 16 + (N + CN) * 4;     // the slab ring + both bias vectors
    static_assert(K % 64 == 0 && N % 128 == 0 && CN % 64 == 0, "shape");
    static_assert(SLAB0 % (NW * 64) == 0 && SLAB1 % (NW * 64) == 0, "whole DMA instructions per slab");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// chunk swizzle of a slab row (as csrc/pw_pair.hip: ds_read_b128 lane groups, MI355X_MICROARCH.md)
template <int CPR> __device__ __forceinline__ int pn_swz(int row) { return CPR >= 16 ? (row & 15) : ((row >> 1) & 7); }

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void pn_wait_vm(int n)
{
    switch (n) {
#define PN_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    PN_W(1) PN_W(2) PN_W(3) PN_W(4) PN_W(5) PN_W(6) PN_W(7) PN_W(8) PN_W(9) PN_W(10) PN_W(11) PN_W(12) PN_W(13) PN_W(14) PN_W(15) PN_W(16)
    PN_W(17) PN_W(18) PN_W(19) PN_W(20) PN_W(21) PN_W(22) PN_W(23) PN_W(24) PN_W(25) PN_W(26) PN_W(27) PN_W(28) PN_W(29) PN_W(30) PN_W(31) PN_W(32)
    PN_W(33) PN_W(34) PN_W(35) PN_W(36) PN_W(37) PN_W(38) PN_W(39) PN_W(40) PN_W(41) PN_W(42) PN_W(43) PN_W(44) PN_W(45) PN_W(46) PN_W(47) PN_W(48)
#undef PN_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Workgroup barrier of the panel kernel.  Leader and trailer waves run two instantiations of the panel loop, so a barrier is a
// DIFFERENT call site in each role: the hardware counts arriving waves whatever their PC, but __syncthreads() in divergent
// control flow is undefined in the HIP model, so the kernel uses the raw instruction and waits for its own LDS operations
// itself.  INVARIANT (checked by tests/test_gpu_ops.py::test_pw_panel_*: a miscount hangs the launch): both roles execute the
// same number of pn_barrier() per panel — 1 (prologue) + G (one per interval) + 1 when a second GEMM's ring must drain.
__device__ __forceinline__ void pn_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int K, int N, int CN, int PB_, bool F16>
__global__ __launch_bounds__(512) void pw_panel_kernel(const PanelK p)
{
    using C = PanelCfg<K, N, CN, PB_>;
    constexpr int PB = C::PB, KS = C::KS, CPR = C::CPR, G = C::G, S = C::S, NB1 = C::NB1;
    extern __shared__ __attribute__((aligned(16))) u32x4 pn_smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    // the biases live in LDS: read in the epilogue they would be vector-memory loads in the middle of the group loop, and
    // vmcnt retires in order — waiting for them would also wait for the residual prefetch of the NEXT group issued before
    float *sBias = (float *)(pn_smem + S * C::SLAB);
    for (int i = tid; i < N; i += 512) sBias[i] = p.bias ? p.bias[i] : 0.0f;
    if constexpr (CN > 0)
        for (int i = tid; i < CN; i += 512) sBias[N + i] = p.b1 ? p.b1[i] : 0.0f;

    // scripts/panel_probe.py: timing builds with parts of the kernel removed (-DUSOT_PNABL_NOMMA / NOSTORE / NORES / NODMA)
    auto dma16 = [&](const uint16_t *src, const u32x4 *dst) {
#ifdef USOT_PNABL_NODMA
        return;
#endif
        const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)dst);
        unsigned keep;
        // asm, not the builtin: beside a builtin LDS-DMA hipcc drains vmcnt before every LDS read (cdna_hip_programming.md)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
    };
    // slabs of group `g` -> ring slot by LDS-DMA.  Physical chunk c = i * 512 + tid of a slab holds logical chunk
    // lc = pc ^ swz(row) of slab row `row` (the swizzle and the row permutation sit on the SOURCE address):
    //   W  slab: row = channel g * 64 + perm(row), all of K;   W1 slab: row = T channel perm(row), k = g * 64 .. g * 64 + 63
    auto issue_slab = [&](int g, int slot) {
        u32x4 *base = pn_smem + slot * C::SLAB;
#pragma unroll
        for (int i = 0; i < C::NI0; ++i) {
            const int c = i * 512 + tid;
            const int row = c / CPR, pc = c % CPR;
            const int lc = pc ^ pn_swz<CPR>(row);
            const int blk = row >> 4, rho = row & 15;
            const int ch = g * 64 + (blk >> 1) * 32 + (rho >> 2) * 8 + (blk & 1) * 4 + (rho & 3);
            dma16(p.w + (long)ch * K + lc * 8, base + i * 512 + wave * 64);
        }
        if constexpr (CN > 0) {
#pragma unroll
            for (int i = 0; i < C::NI1; ++i) {
                const int c = i * 512 + tid;
                const int row = c / 8, pc = c % 8;
                const int lc = pc ^ pn_swz<8>(row);
                const int blk = row >> 4, rho = row & 15;                  // blk = 16-channel block of T, four per 64
                const int ch = (blk >> 2) * 64 + ((blk >> 1) & 1) * 32 + (rho >> 2) * 8 + (blk & 1) * 4 + (rho & 3);
                dma16(p.w1 + (long)ch * N + g * 64 + lc * 8, base + C::SLAB0 + i * 512 + wave * 64);
            }
        }
    };

    // Waves 0-3 LEAD, waves 4-7 TRAIL by half an interval (a workgroup's waves are dealt round-robin to the four SIMDs, so
    // every SIMD hosts one of each).  Between two barriers a leader runs [GEMM(k), epilogue(k)], a trailer [epilogue(k-1),
    // GEMM(k)]: while one wave of a SIMD issues MFMAs the other runs its epilogue (VALU, residual, stores).  In lock step
    // the two phases of both waves were serial: with the residual, the stores and the DMA all compiled out the kernel still
    // took 34 us of the launch's 65 (scripts/panel_probe.py).
    auto run = [&](auto role) {
    constexpr bool trail = decltype(role)::value;              // two fully specialised instruction streams
    for (int panel = blockIdx.x; panel < p.npanels; panel += gridDim.x) {
        const long pm0 = (long)panel * C::BM + wave * (PB * 16);
        const bool full = pm0 + PB * 16 <= (long)p.M;            // wave-uniform: every pixel row of this wave exists
        // ring of three slots, slab k in slot k % 3: during interval k leaders read slab k, trailers slab k-1 (second GEMM)
        // and slab k, and the DMA fills slot (k+1) % 3, which held slab k-2: free since the barrier that ended interval k-1
        issue_slab(0, 0);
        // ---- this wave's B operands: X fragments of its PB pixel blocks, all of K, straight into registers
        u32x4 xf[PB][KS];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const long m = min(pm0 + j * 16 + l15, (long)p.M - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[j][ks] = *(const u32x4 *)(p.x + m * K + ks * 32 + q * 8);
        }
        // residual of the lane's two 8-channel runs (c0 = g * 64 + q * 8 and c0 + 32), PB pixel rows, loaded a whole interval
        // before its epilogue (two register sets, the interval loop is unrolled by two so that they are static)
        u32x4 ra[PB][2], rb[PB][2];
        auto load_res = [&](int g, u32x4 (&rr)[PB][2]) {
            if (!p.res) return;
#ifdef USOT_PNABL_NORES
            for (int j = 0; j < PB; ++j) rr[j][0] = rr[j][1] = u32x4{(uint32_t)g, 1u, 2u, 3u};
            return;
#endif
            const int c0 = min(g, G - 1) * 64 + q * 8;        // past the last group: a harmless re-read keeps the op count fixed
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const long m = min(pm0 + j * 16 + l15, (long)p.M - 1);
                rr[j][0] = *(const u32x4 *)(p.res + m * N + c0);
                rr[j][1] = *(const u32x4 *)(p.res + m * N + c0 + 32);
            }
        };
        if constexpr (!trail) load_res(0, ra);
        f32x4 acc[4][PB];                                     // group accumulators: live from a GEMM to its epilogue
        f32x4 acct[NB1 ? NB1 : 1][PB];                        // the second GEMM's accumulators live across the groups
#pragma unroll
        for (int n = 0; n < (NB1 ? NB1 : 1); ++n)
#pragma unroll
            for (int j = 0; j < PB; ++j) acct[n][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the DMA is invisible to the compiler's wait counts
        pn_barrier();

        auto gemm = [&](int g) {
            const u32x4 *slab = pn_smem + (g % 3) * C::SLAB;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // A fragments through a ring of WD k-steps, pinned: hipcc otherwise reads two fragments, waits, issues two MFMAs,
            // waits again — the LDS latency exposed every 34 MFMA cycles (with everything but the MFMAs compiled out the launch
            // still took 34 us of 65).  With the reads of k-step ks + WD - 1 issued BEFORE the MFMAs of k-step ks the matrix
            // pipe only ever waits for reads issued a whole k-step earlier (ds_read returns in order: counted lgkmcnt).
            constexpr int WD = (KS >= 4 && CN == 0) ? 3 : 2;     // the pair form has no registers to spare
            u32x4 wf[WD][4];
            auto read_w = [&](int ks, u32x4 (&w)[4]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 16 + l15;
                    w[i] = slab[row * CPR + ((ks * 4 + q) ^ pn_swz<CPR>(row))];
                }
            };
#pragma unroll
            for (int d = 0; d < WD - 1; ++d) read_w(d, wf[d]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + WD - 1 < KS) read_w(ks + WD - 1, wf[(ks + WD - 1) % WD]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
#ifndef USOT_PNABL_NOMMA
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < PB; ++j) acc[i][j] = pn_mfma<F16>(wf[ks % WD][i], xf[j][ks], acc[i][j]);
#else
                for (int i = 0; i < 4; ++i) acc[i][0][0] += __builtin_bit_cast(float, wf[ks % WD][i][0] ^ xf[0][ks][1]);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // epilogue of group g from registers: acc[i][j][r] = channel c0 + (i >> 1) * 32 + (i & 1) * 4 + r of pixel row j * 16 + l15
        auto epilogue = [&](int g, u32x4 (&rr)[PB][2]) {
            const int c0 = g * 64 + q * 8;                    // the lane's two 8-channel runs: c0 .. c0+7 and c0+32 .. c0+39
            f32x4 bias[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bias[i] = *(const f32x4 *)(sBias + c0 + (i >> 1) * 32 + (i & 1) * 4);
            u32x4 yf[PB][2];                                  // the lane's 16 channels in the storage type
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const long m = pm0 + j * 16 + l15;
                float v[16];                                      // v[e]: run e / 8 (= block i >> 1), element e % 8 of it
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = i * 4 + r;
                        v[e] = acc[i][j][r] + bias[i][r];
                        if (p.res) {
                            const uint32_t rw = rr[j][e / 8][(e % 8) / 2];
                            v[e] += pn_unpack<F16>((e & 1) ? (rw >> 16) : (rw & 0xffffu));
                        }
                        if (p.act == USOT_ACT_RELU) v[e] = fmaxf(v[e], 0.0f);
                    }
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) yf[j][k][e] = usot_pack2_lp<F16>(v[k * 8 + 2 * e], v[k * 8 + 2 * e + 1]);
#ifdef USOT_PNABL_NOSTORE
                if (yf[j][0][0] == 0x12345678u && yf[j][1][3] == 0x9abcdef0u)      // keeps the values live, never true in practice
#else
                if (full || m < p.M)
#endif
                {
                    *(u32x4 *)(p.y + m * N + c0) = yf[j][0];
                    *(u32x4 *)(p.y + m * N + c0 + 32) = yf[j][1];
                }
            }
            // ---- second GEMM, this group's 64 k: the lane's rounded outputs are its B fragments (k-step s = run s)
            if constexpr (CN > 0) {
                const u32x4 *slab1 = pn_smem + (g % 3) * C::SLAB + C::SLAB0;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int n = 0; n < NB1; ++n) {
                        const int row = n * 16 + l15;
                        const u32x4 w1f = slab1[row * 8 + ((s * 4 + q) ^ pn_swz<8>(row))];
#pragma unroll
                        for (int j = 0; j < PB; ++j) acct[n][j] = pn_mfma<F16>(w1f, yf[j][s], acct[n][j]);
                    }
            }
        };
        // one interval; rc / rn: the residual register sets of the epilogue run in it / of the prefetch issued in it
        auto interval = [&](int k, u32x4 (&rc)[PB][2], u32x4 (&rn)[PB][2]) {
            if (k + 1 < G) issue_slab(k + 1, (k + 1) % 3);
            bool stored = true;
            if constexpr (!trail) {
                load_res(k + 1, rn);
                gemm(k);
                epilogue(k, rc);
            } else {
                load_res(k, rn);                              // consumed at the top of the NEXT interval
                if (k > 0) epilogue(k - 1, rc);
                else stored = false;
                gemm(k);
            }
            // Before the barrier slab k+1 must have landed: its DMAs were issued at the top of this interval, so it is enough
            // that at most the vector-memory operations issued AFTER them are outstanding — this interval's residual prefetch
            // (2 PB loads, if any) and stores (2 PB).  A wave of the ragged last panel may have skipped stores: it drains.
            if (!full) pn_wait_vm(0);
            else       pn_wait_vm((p.res ? 2 * PB : 0) + (stored ? 2 * PB : 0));
            pn_barrier();
        };
        static_assert(G % 2 == 0 && S == 3, "interval loop unrolled by two over a ring of three slabs");
        // leaders: interval k runs epilogue(k) on the set loaded in interval k-1 (k even: ra) and loads k+1 into the other;
        // trailers: interval k runs epilogue(k-1) on the set loaded in interval k-1 (k-1 even: ra ... ) and loads k
#pragma unroll 1
        for (int k = 0; k < G; k += 2) {
            if constexpr (!trail) { interval(k, ra, rb); interval(k + 1, rb, ra); }
            else                  { interval(k, rb, ra); interval(k + 1, ra, rb); }
        }
        if constexpr (trail) epilogue(G - 1, rb);
        // ---- epilogue of the second GEMM: acct[n][j][r] = T channel (n >> 2) * 64 + ((n >> 1) & 1) * 32 + q * 8 + (n & 1) * 4 + r
        if constexpr (CN > 0) {
#pragma unroll
            for (int nb = 0; nb < NB1 / 4; ++nb) {
                const int t0 = nb * 64 + q * 8;
                f32x4 b1v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) b1v[i] = *(const f32x4 *)(sBias + N + t0 + (i >> 1) * 32 + (i & 1) * 4);
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const long m = pm0 + j * 16 + l15;
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[i * 4 + r] = acct[nb * 4 + i][j][r] + b1v[i][r];
                            if (p.act2 == USOT_ACT_RELU) v[i * 4 + r] = fmaxf(v[i * 4 + r], 0.0f);
                        }
                    if (m < p.M) {
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            u32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = usot_pack2_lp<F16>(v[k * 8 + 2 * e], v[k * 8 + 2 * e + 1]);
                            *(u32x4 *)(p.t + m * CN + t0 + k * 32) = o;
                        }
                    }
                }
            }
        }
        // the next panel's prologue refills slot 0: every wave must be done with the ring (the trailers' last epilogue reads
        // the second GEMM's slab after the loop's last barrier)
        if (CN > 0 && panel + (int)gridDim.x < p.npanels) pn_barrier();
    }
    };
    if (wave >= 4) run(std::true_type{});
    else           run(std::false_type{});
}

template <int K, int N, int CN, int PB>
int panel_launch(hipStream_t s, PanelK p, int dtype)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    using C = PanelCfg<K, N, CN, PB>;
    static bool raised_d[USOT_MAX_DEV][2] = {};
    bool (&raised)[2] = raised_d[usot_dv];
    const void *fn = dtype ? (const void *)pw_panel_kernel<K, N, CN, PB, true> : (const void *)pw_panel_kernel<K, N, CN, PB, false>;
    if (C::LDS_BYTES > 64 * 1024 && !raised[dtype]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) return USOT_ELAUNCH;
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
    p.npanels = (p.M + C::BM - 1) / C::BM;
    const int grid = p.npanels < cus ? p.npanels : cus;        // persistent: one 8-wave workgroup per CU (up to 256 registers)
    if (dtype) hipLaunchKernelGGL((pw_panel_kernel<K, N, CN, PB, true>), dim3(grid), dim3(512), C::LDS_BYTES, s, p);
    else       hipLaunchKernelGGL((pw_panel_kernel<K, N, CN, PB, false>), dim3(grid), dim3(512), C::LDS_BYTES, s, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

}  // namespace

extern "C" int usot_pw_panel_pair_supported(int CM, int CO, int CN);

extern "C" int usot_pw_panel_supported(int K, int N)
{
    return (K == 256 && N == 1024) || (K == 128 && N == 512) || (K == 64 && N == 256);
}

/* y[M][N] = act(x[M][K] . w^T + bias (+ res)): x, w ([N][K], the conv kernels' layout), res, y in the storage type
 * (dtype 0 = bf16, 1 = fp16), bias fp32 or NULL, res NULL or [M][N]; act USOT_ACT_NONE | USOT_ACT_RELU. */
extern "C" int usot_pw_panel_lp(void *stream, const void *x, const void *w, const float *bias, const void *res, void *y,
                                int M, int K, int N, int act, int dtype)
{
    if (!x || !w || !y || M <= 0 || (dtype != 0 && dtype != 1) || !usot_pw_panel_supported(K, N)) return USOT_EINVAL;
    if (act != USOT_ACT_NONE && act != USOT_ACT_RELU) return USOT_EINVAL;
    const void *ptrs[] = {x, w, bias, res, y};
    for (const void *q : ptrs)
        if ((uintptr_t)q % 16) return USOT_EINVAL;
    PanelK p = {};
    p.x = (const uint16_t *)x; p.w = (const uint16_t *)w; p.res = (const uint16_t *)res; p.bias = bias; p.y = (uint16_t *)y;
    p.M = M; p.act = act;
    hipStream_t s = (hipStream_t)stream;
    // fewer than 192 panels of 256 pixels (batch 32 at layer2 / layer3 resolution: 121) would leave half the chip idle: panels of
    // 128 pixels then (one pixel block per wave; round 4)
    const bool small = (M + 255) / 256 < 192;
    if (K == 256) return small ? panel_launch<256, 1024, 0, 1>(s, p, dtype) : panel_launch<256, 1024, 0, 2>(s, p, dtype);
    if (K == 128) return small ? panel_launch<128, 512, 0, 1>(s, p, dtype) : panel_launch<128, 512, 0, 2>(s, p, dtype);
    return panel_launch<64, 256, 0, 4>(s, p, dtype);
}

/* pixels per panel (= per workgroup) of a supported shape, CN = 0 for the single convolution; 0 = unsupported.  A launch
 * keeps ceil(M / pixels) of the CUs busy: callers should prefer the tiled kernels when that is far below the CU count. */
extern "C" int usot_pw_panel_pixels(int CM, int CO, int CN)
{
    if (CN == 0) {
        if (!usot_pw_panel_supported(CM, CO)) return 0;
        return CM == 256 ? PanelCfg<256, 1024, 0, 2>::BM : CM == 128 ? PanelCfg<128, 512, 0, 2>::BM : PanelCfg<64, 256, 0, 4>::BM;
    }
    if (!usot_pw_panel_pair_supported(CM, CO, CN)) return 0;
    if (CM == 128) return CN == 128 ? PanelCfg<128, 512, 128, 2>::BM : PanelCfg<128, 512, 256, 1>::BM;
    return CN == 64 ? PanelCfg<64, 256, 64, 2>::BM : PanelCfg<64, 256, 128, 2>::BM;
}

/* the smallest panel the single convolution runs on (the launcher switches to it below 192 panels of the default size) */
extern "C" int usot_pw_panel_min_pixels(int K, int N)
{
    if (!usot_pw_panel_supported(K, N)) return 0;
    return K == 64 ? PanelCfg<64, 256, 0, 4>::BM : 128;
}

extern "C" int usot_pw_panel_pair_supported(int CM, int CO, int CN)
{
    // (layer3's (256, 1024, 256) would need one pixel block per wave — every W / W1 fragment read from LDS feeds ONE MFMA — and
    //  measured 122 us against 66 + 47 for the panel conv3 and the tiled conv1; its two slabs per group do not fit a ring of three)
    return (CM == 128 && CO == 512 && CN == 128) || (CM == 128 && CO == 512 && CN == 256) ||
           (CM == 64 && CO == 256 && CN == 64) || (CM == 64 && CO == 256 && CN == 128);
}

/* The fused pair in the pixel-stationary form: Y = relu(t2 . w3^T + b3 + res) (stored), T = act2(Y . w1^T + b1).  Descriptor
 * as for usot_pw_pair_lp EXCEPT that both filter banks are in the conv kernels' natural layout: d->w3p = w3 [CO][CM],
 * d->w1 = w1 [CN][CO].  Same numerics as the two launches: fp32 accumulation in the same k order, Y rounded once. */
extern "C" int usot_pw_panel_pair_lp(void *stream, const usot_pw_pair_desc *d, int dtype)
{
    if (!d || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    if (!d->t2 || !d->w3p || !d->b3 || !d->res || !d->y || !d->w1 || !d->b1 || !d->t || d->M <= 0) return USOT_EINVAL;
    if (d->act2 != USOT_ACT_NONE && d->act2 != USOT_ACT_RELU) return USOT_EINVAL;
    if (!usot_pw_panel_pair_supported(d->CM, d->CO, d->CN)) return USOT_EINVAL;
    const void *ptrs[] = {d->t2, d->w3p, d->b3, d->res, d->y, d->w1, d->b1, d->t};
    for (const void *q : ptrs)
        if ((uintptr_t)q % 16) return USOT_EINVAL;
    PanelK p = {};
    p.x = (const uint16_t *)d->t2; p.w = (const uint16_t *)d->w3p; p.res = (const uint16_t *)d->res; p.w1 = (const uint16_t *)d->w1;
    p.bias = d->b3; p.b1 = d->b1; p.y = (uint16_t *)d->y; p.t = (uint16_t *)d->t;
    p.M = d->M; p.act = USOT_ACT_RELU; p.act2 = d->act2;
    hipStream_t s = (hipStream_t)stream;
    if (d->CM == 128 && d->CN == 128) return panel_launch<128, 512, 128, 2>(s, p, dtype);
    if (d->CM == 128) return panel_launch<128, 512, 256, 1>(s, p, dtype);
    if (d->CN == 64) return panel_launch<64, 256, 64, 2>(s, p, dtype);
    return panel_launch<64, 256, 128, 2>(s, p, dtype);
}
