// A layer3 bottleneck of the batched low-precision backbone (conv2 3x3 -> conv3 1x1 + residual + ReLU -> the NEXT block's conv1
// 1x1; modules.py:43-56, 40-42) in ONE launch whose matrix-pipe work and HBM work OVERLAP on every CU.
//
// csrc/conv_pw_lp.hip runs the same three convolutions per 256-pixel panel as phases of ONE 16-wave workgroup per CU: ~62 us of
// matrix pipe (conv2), then ~50 us of HBM (residual in, Y out, conv3's MFMAs hidden under it), then ~38 us of conv1 on the Y panel
// read back - in sequence on every CU, 149 us per block, 6 blocks = 44 % of the batch-64 bf16 step.  A phase cannot overlap
// the next one of the SAME panel (conv3 needs all of conv2's k); it can overlap another panel's.  So here the two kinds of work
// belong to two kinds of workgroup, 8 wavefronts / <= 128 registers / 72 KB of LDS each, so that every CU hosts one of each:
//
//   M ("matrix") workgroup   P1(panel): conv2 of a 128-pixel panel, 128 x 256 x 32 k-tiles through a ring of three LDS stages filled by
//                            LDS-DMA (row-shared: one activation stage per (kh, 32-channel chunk) serves the three kw taps), the T2
//                            panel leaves through L2 (write-through 8-byte stores + one flag);
//                            P5(panel): the next conv1 (1024 -> 256) on the panel's Y rows, once its H partner has published them.
//   H ("HBM") workgroup      P4(panel): waits for the T2 panel, takes its wave's 16 pixels x 256 k into registers, then conv3 group by
//                            group exactly as csrc/conv_pw_lp.hip's phase 4 (w3 slabs through a ring of two LDS slots, residual one
//                            group ahead, 16-byte register epilogue), Y written through, one flag.
//
// A pair (M, H) walks the same panels p, p + G, ...: M runs P1(p0), P1(p1), P5(p0), P5(p1), H runs P4(p0), P4(p1) - while H streams
// panel p0's residual and Y, M's MFMAs work on p1; while H streams p1, M runs conv1 on p0.  Hand-offs are placement-independent
// (cdna_hip_programming.md Guideline 16, recipe R1 / MI355X_MICROARCH.md "publish-large"): write-through (sc1) payload stores, every
// wave drains vmcnt, workgroup barrier, ONE relaxed agent-scope flag store; the consumer polls the flag relaxed from one lane
// (s_sleep between polls, bounded), then one agent-scope acquire, barrier, plain loads.  Flags are reset by their consumer: every
// launch - and every graph replay - finds them zero.  Which CU a workgroup lands on is a matter of speed only: blocks b and b + 8 of
// a pair share an XCD (b % 8), and the role pattern alternates so that the dispatcher's round-robin over an XCD's 32 CUs gives each
// CU one M and one H.
//
// MEASURED (round 6, scripts/conv_pw_ov_probe.py --trace; batch 64, bf16): parity-green, deterministic, flags replay-safe - and SLOWER:
// 234 us per block against 146 us for the sequential phases of csrc/conv_pw_lp.hip.  The trace (240 of 256 CUs host one M and one H,
// as designed) says why: a 128 x 256 x 32 k-tile takes 1 575 cycles for its 512 cycles of MFMAs when M runs alone (P1 of the first
// panel: 54 us; the 256-pixel panel of the sequential kernel needs 62), 2 250 beside a streaming H (78 us), and H's residual / Y phase
// stretches from 25 to 57-60 us beside M's LDS-DMA traffic: the two workgroups do not complement each other, they queue on the same
// vector-memory path (scripts/probes/fill_probe.hip: a CU fills LDS from L2 at 44 B/clk, from the Infinity Cache at 24 B/clk, through
// LDS-DMA and through registers alike), and the smaller tile the 72 KB / 128-register budget forces has half the FLOPs per byte moved
// into LDS.  EXPERIMENT: compiled with -DUSOT_EXPERIMENTS only, engine option conv_pw_ov_lp (default off).
//
// Arithmetic: as conv_pw_lp.hip with the row-shared k-loop, k order (kh, 32-channel chunk, kw): same products, fp32 accumulation,
// T2 and Y rounded to the storage type where that kernel rounds them; conv1 in k order like the tiled kernels (bit-identical to
// usot_conv2d_lp on this kernel's Y).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "usot_hip.h"
#include "common.h"

#ifdef USOT_EXPERIMENTS
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct OvK {
    const uint16_t *x, *w2, *w3, *res, *zero, *w1;
    const float *b2, *b3, *b1;
    uint16_t *y, *t, *t2s;             // t2s: scratch map [M][256] in the storage type (the T2 hand-off)
    int *flags;                        // [2 NP]: t2_ready[p], y_ready[NP + p]; zero before the first launch, left zero by every launch
    int *err;                          // sticky: a bounded flag wait ran out (the launch then finishes on garbage instead of hanging)
    long long *dbg;                    // nullptr, or 32 slots per block: role, XCC id, HW id, then (phase tag, 100 MHz time stamp) pairs
    int H, W, OW, pad_h, dil_h, dil_w;
    int M, P, NP, G8, R, act2;
};

__device__ __attribute__((aligned(16))) uint32_t ov_zero16[4] = {0u, 0u, 0u, 0u};

constexpr int OV_BM = 128, OV_CM = 256, OV_CO = 1024, OV_CN = 256, OV_NW = 8, OV_NTHR = 512;
// LDS (16-byte chunks).  P1: two activation stages of 144 rows x 4 chunks, three filter stages of 256 x 4, one row of zeros.
constexpr int OV_XST = 144 * 4, OV_WST = 256 * 4, OV_WB = 2 * OV_XST, OV_ZB = OV_WB + 3 * OV_WST;
// P5: three Y stages of 128 x 4, three filter stages of 256 x 4.  H: two w3 slabs of 64 x 32 chunks, then the biases.
constexpr int OV_YST = 128 * 4, OV_W1B = 3 * OV_YST;
constexpr int OV_SLAB = 64 * 32;
constexpr int OV_XTAB = OV_ZB + 4;                           // P1: per staged activation row (144): element offset of its pixel, its oh - pad
constexpr int OV_LDS_P1 = (OV_XTAB + 144 / 2) * 16, OV_LDS_P5 = (OV_W1B + 3 * OV_WST) * 16, OV_LDS_H = 2 * OV_SLAB * 16 + (OV_CO + OV_CN) * 4;
constexpr int OV_LDS_T = OV_BM * (OV_CN / 8) * 16;            // P5's output tile on its way out
constexpr int ov_max(int a, int b) { return a > b ? a : b; }
constexpr int OV_LDS = ov_max(ov_max(OV_LDS_P1, OV_LDS_P5), ov_max(OV_LDS_H, OV_LDS_T));
static_assert(2 * OV_LDS <= 160 * 1024, "two workgroups per CU");

template <bool F16> __device__ __forceinline__ f32x4 ov_mfma(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else               return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ float ov_unpack(uint32_t h)
{
    return F16 ? (float)__builtin_bit_cast(_Float16, (uint16_t)h) : __builtin_bit_cast(float, h << 16);
}
// staged rows are 64 bytes (32 k): chunk c of row r sits at chunk c ^ ((r >> 1) & 3) - a ds_read_b128 lane group (16 rows x one
// chunk, MI355X_MICROARCH.md LDS table) then covers the sixteen 16-byte slots of a bank row whatever the first row is
__device__ __forceinline__ int ov_swz(int row) { return (row >> 1) & 3; }

__device__ __forceinline__ void ov_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// one lane polls a flag (relaxed, agent scope; bounded), the workgroup then acquires
__device__ __forceinline__ void ov_wait_flag(int *flag, int *err, int tid)
{
    if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1 << 21)) {                     // seconds: the partner never ran - say so and go on rather than hang the GPU
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // consumed: the next launch finds it zero
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__device__ __forceinline__ void ov_publish(int *flag, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every wave: its write-through stores have left
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool F16>
__global__ __launch_bounds__(OV_NTHR, 4) void conv_pw_ov_kernel(const OvK p)
{
    constexpr int BM = OV_BM, CM = OV_CM, CO = OV_CO, CN = OV_CN, NW = OV_NW, NTHR = OV_NTHR;
    extern __shared__ __attribute__((aligned(16))) u32x4 ov_smem[];
    const int tid_ = threadIdx.x, lane_ = tid_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    // scripts/conv_pw_ov_probe.py --trace: where each workgroup ran and when each of its phases began and ended
    int dbg_n = 3;
    auto stamp = [&](int tag) {
        if (p.dbg && tid_ == 0 && dbg_n < 31) {
            long long *d = p.dbg + (long)blockIdx.x * 32;
            d[dbg_n++] = ((long long)tag << 56) | (long long)(__builtin_amdgcn_s_memrealtime() & 0xffffffffffffffLL);
        }
    };
    // block -> (XCD, index on it) -> (pair, role).  Consecutive blocks of an XCD alternate M / H, and the pattern flips every 32 blocks
    // (an XCD's CU count) so that a CU's second workgroup is of the other kind than its first; pair = the two neighbours (2 k, 2 k + 1)
    const int xcd = (int)blockIdx.x & 7, bi = (int)blockIdx.x >> 3;
    const bool role_h = ((bi + (bi >> 5)) & 1) != 0;
    const int per_xcd = p.G8 >> 3;                          // pairs per XCD
    const int pl = bi >> 1;                                 // pair index on this XCD
    if (pl >= per_xcd) return;
    if (p.dbg && tid_ == 0) {
        long long *d = p.dbg + (long)blockIdx.x * 32;
        d[0] = role_h ? 1 : 0;
        d[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        d[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: wave, SIMD, pipe, CU, SH, SE
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)ov_smem;
    auto dma16u = [&](const uint16_t *src, uint32_t lds) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
    };
    // the same with a wave-uniform base pointer in SGPRs and a 32-bit byte offset per lane (one address register instead of two)
    auto dma16s = [&](const uint16_t *base, uint32_t off, uint32_t lds) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(lds) : "memory");
    };
    // round r: panels [r G8, (r + 1) G8), an XCD's pairs take a contiguous run of them (neighbouring panels share halo rows)
    auto panel_of = [&](int r) { return r * p.G8 + xcd * per_xcd + pl; };
    const __amdgpu_buffer_rsrc_t t2rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.t2s, 0, (int)((long)p.M * CM * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.res, 0, (int)((long)p.M * CO * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, (int)((long)p.M * CO * 2), 0x00020000);

    if (!role_h) {
        // ============================================================== M: conv2 (P1) and the next conv1 (P5)
        const int wm = wave & 1, wn = wave >> 1;           // 2 x 4 waves over 128 pixels x 256 channels: a wave = 64 x 64
        auto p1 = [&](int panel) {
            // (everything a lane derives from its index is recomputed per call: hoisted out of the panel loop by the compiler, the
            //  store / fragment offsets of P1 and P5 together do not fit the 128 registers two co-resident workgroups leave each
            //  other, and a spill is a scratch access on the vmcnt queue the DMA pipeline counts by hand)
            int lane = lane_, tid = tid_;
            asm volatile("" : "+v"(lane), "+v"(tid));
            const int l15 = lane & 15, quad = lane >> 4;
            const int bm0 = panel * BM;
            f32x4 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (tid < 4) ov_smem[OV_ZB + tid] = u32x4{0u, 0u, 0u, 0u};
            // the staged activation rows (144: pixels bm0 - 4 ... bm0 + 139 at their kh row): element offset of the pixel's column in
            // its image and oh - pad, once per panel, in LDS (six registers per lane otherwise, in a loop that has none to spare)
            int *xtab = (int *)(ov_smem + OV_XTAB);
            if (tid < 144) {
                const int v = bm0 - 4 + tid;
                const bool ok = v >= 0 && v < p.M;
                const int vv = ok ? v : 0;
                const int n = vv / p.P, pix = vv - n * p.P;
                const int oh = pix / p.OW, ow = pix - oh * p.OW;
                xtab[2 * tid] = (n * p.H * p.W + ow) * CM;
                xtab[2 * tid + 1] = ok ? oh - p.pad_h : -(1 << 20);
            }
            __syncthreads();
            // activation share: waves 0-2 move one 16-row block per k-tile: block kw * 3 + wave of the next group's stage (a lane =
            // (row of the block, 16-byte chunk))
            auto issue_x = [&](int kw, int g, int stage) {
                if (wave < 3) {
                    const int row = (kw * 3 + wave) * 16 + (lane >> 2);
                    const int xo = xtab[2 * row], xh = xtab[2 * row + 1];
                    const int kh = g >> 3, cc = g & 7;
                    const int ih = xh + kh * p.dil_h;
                    const bool ok = (unsigned)ih < (unsigned)p.H;
                    dma16u(ok ? p.x + ((uint32_t)xo + (uint32_t)(ih * p.W * CM + cc * 32 + ((lane & 3) ^ ov_swz(row)) * 8)) : p.zero,
                           __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((stage * OV_XST + (kw * 3 + wave) * 64) * 16)));
                }
            };
            // filter k-tile: 256 rows x 64 bytes = sixteen 16-row blocks, two per wave (rows 16 apart share their swizzle)
            constexpr uint32_t K2 = 9u * CM;
            const int wrow = wave * 32 + (lane >> 2);
            const uint32_t w_off = ((uint32_t)wrow * K2 + (uint32_t)(((lane & 3) ^ ov_swz(wrow)) * 8)) * 2u;      // bytes
            const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((OV_WB + wave * 2 * 64) * 16));
            auto issue_w = [&](int t) {                      // k-tile t = 3 g + kw -> filter slot t % 3
                const int g = t / 3, kw = t - g * 3;
                const int kh = g >> 3, cc = g & 7;
                const uint32_t koff = (uint32_t)((kh * 3 + kw) * CM + cc * 32) * 2u;
                const uint32_t dst = ldsw + (uint32_t)((t % 3) * OV_WST * 16);
                dma16s(p.w2, w_off + koff, dst);
                dma16s(p.w2, w_off + koff + 16u * K2 * 2u, dst + 64 * 16);
            };
            uint32_t tapmask = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ow = (bm0 + wm * 64 + j * 16 + l15) % p.OW;
                tapmask |= (ow - p.dil_w >= 0 ? 1u : 0u) << j;
                tapmask |= (ow + p.dil_w < p.W ? 1u : 0u) << (4 + j);
            }
            const int row0 = 4 + wm * 64 + l15;
            constexpr int NG = 24, NT = 72;
            using K0 = std::integral_constant<int, 0>;
            using K1 = std::integral_constant<int, 1>;
            using K2c = std::integral_constant<int, 2>;
            issue_x(0, 0, 0); issue_x(1, 0, 0); issue_x(2, 0, 0);
            issue_w(0);
            issue_w(1);
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");     // all but k-tile 1's filters
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int swl = ov_swz(l15);
            auto ktile = [&](auto kwc, int g) {
                constexpr int kw = decltype(kwc)::value;
                const int t = g * 3 + kw;
                // this k-tile's share of the next group's activation stage FIRST, then the filters of k-tile t + 2: the wait below
                // leaves exactly those two filter pieces in flight
                if (g + 1 < NG) issue_x(kw, g + 1, (g + 1) & 1);
                if (t + 2 < NT) issue_w(t + 2);
                int sh = (kw - 1) * p.dil_w;
                asm volatile("" : "+s"(sh));
                u32x4 wf[4], xf[4];
                const u32x4 *cW = ov_smem + OV_WB + (t % 3) * OV_WST + (wn * 64 + l15) * 4 + (quad ^ swl);
                // rows 16 apart share their swizzle: one address per k-tile, the four blocks at fixed distances
                const int xrow = row0 + sh;
                const int xa = (g & 1) * OV_XST + xrow * 4 + (quad ^ ov_swz(xrow));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool in = kw == 1 || ((tapmask >> ((kw == 0 ? 0 : 4) + j)) & 1u);
                    xf[j] = ov_smem[in ? xa + j * 64 : OV_ZB + quad];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = cW[i * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = ov_mfma<F16>(wf[i], xf[j], acc[i][j]);
                if (t + 2 < NT) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                else            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            };
#pragma unroll 1
            for (int g = 0; g < NG; ++g) {
                ktile(K0{}, g);
                ktile(K1{}, g);
                ktile(K2c{}, g);
            }
            // T2 = relu(acc + b2), rounded to the storage type, written through to the scratch map: a lane owns 4 consecutive channels
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = wn * 64 + i * 16 + quad * 4;
                const f32x4 b = *(const f32x4 *)(p.b2 + ch);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long m = (long)bm0 + wm * 64 + j * 16 + l15;
                    f32x4 v = acc[i][j] + b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                    u32x2 o;
                    o[0] = usot_pack2_lp<F16>(v[0], v[1]);
                    o[1] = usot_pack2_lp<F16>(v[2], v[3]);
                    if (m < (long)p.M) __builtin_amdgcn_raw_buffer_store_b64(o, t2rs, (int)((m * CM + ch) * 2), 0, 16);
                }
            }
            ov_publish(p.flags + panel, tid);
        };
        auto p5 = [&](int panel) {
            // T[BM][CN] = act2(Y . w1^T + b1), K = 1024 in 32 k-tiles of 32: Y rows (published by the H partner) two k-tiles ahead,
            // filters likewise; a wave = 64 pixels x 64 channels
            int lane = lane_, tid = tid_;
            asm volatile("" : "+v"(lane), "+v"(tid));
            const int l15 = lane & 15, quad = lane >> 4;
            const int bm0 = panel * BM;
            ov_wait_flag(p.flags + p.NP + panel, p.err, tid);
            stamp(4);
            f32x4 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // Y stage: 128 rows x 64 bytes = eight 16-row blocks, one per wave; filters: sixteen, two per wave
            // (rows past the last pixel read the last pixel's: their results are never stored; byte offsets fit 32 bits: launcher)
            const int yrow = wave * 16 + (lane >> 2);
            const long ym = (long)bm0 + yrow;
            const uint32_t y_off = (uint32_t)(((ym < (long)p.M ? ym : (long)p.M - 1) * CO + ((lane & 3) ^ ov_swz(yrow)) * 8) * 2);
            const int w1row = wave * 32 + (lane >> 2);
            const uint32_t w1_off = (uint32_t)((w1row * CO + ((lane & 3) ^ ov_swz(w1row)) * 8) * 2);
            const uint32_t ldsy = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(wave * 64 * 16));
            const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((OV_W1B + wave * 2 * 64) * 16));
            auto issue = [&](int t) {                         // one Y piece, then two filter pieces
                const int s = t % 3;
                dma16s(p.y, y_off + (uint32_t)(t * 64), ldsy + (uint32_t)(s * OV_YST * 16));
                dma16s(p.w1, w1_off + (uint32_t)(t * 64), ldsw + (uint32_t)(s * OV_WST * 16));
                dma16s(p.w1, w1_off + (uint32_t)(t * 64 + 16 * CO * 2), ldsw + (uint32_t)(s * OV_WST * 16) + 64 * 16);
            };
            constexpr int NT = CO / 32;
            issue(0);
            issue(1);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int swl = ov_swz(l15);
#pragma unroll 1
            for (int t = 0; t < NT; ++t) {
                if (t + 2 < NT) issue(t + 2);
                const int s = t % 3;
                const u32x4 *cX = ov_smem + s * OV_YST + (wm * 64 + l15) * 4 + (quad ^ swl);
                const u32x4 *cW = ov_smem + OV_W1B + s * OV_WST + (wn * 64 + l15) * 4 + (quad ^ swl);
                u32x4 wf[4], xf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = cW[i * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[j] = cX[j * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = ov_mfma<F16>(wf[i], xf[j], acc[i][j]);
                if (t + 2 < NT) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
                else            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // T tile -> LDS in the storage type (rows of 32 chunks, chunk c of row r at c ^ (r & 15)), then out in whole rows
            char *sT = (char *)ov_smem;
            constexpr int TCPR = CN / 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = wn * 64 + i * 16 + quad * 4;
                const f32x4 b = *(const f32x4 *)(p.b1 + ch);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = wm * 64 + j * 16 + l15;
                    f32x4 v = acc[i][j] + b;
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
                if (m < (long)p.M) *(u32x4 *)(p.t + m * CN + (pc ^ (row & 15)) * 8) = ov_smem[idx];
            }
            __syncthreads();                                  // the LDS is free for the next phase
        };
        // P1 runs at most two panels ahead of P5: P1(p0), P1(p1), P5(p0), P1(p2), P5(p1), ... - while the H partner streams panel i's
        // residual and Y, this workgroup's MFMAs are on panel i + 1's conv2 or on panel i - 1's conv1
        int nvalid = 0;
        for (int r = 0; r < p.R; ++r) nvalid += panel_of(r) < p.NP ? 1 : 0;
        int i1 = 0, i5 = 0;
#pragma unroll 1
        while (i5 < nvalid) {
            if (i1 < nvalid && i1 - i5 < 2) { stamp(1); p1(panel_of(i1)); stamp(2); ++i1; }
            else                            { stamp(3); p5(panel_of(i5)); stamp(6); ++i5; }
        }
        return;
    }

    // ================================================================== H: conv3 + residual + ReLU over the partner's T2 panels
    constexpr int CPR = CM / 8, KS = CM / 32, GR = CO / 64;
    float *sBias = (float *)(ov_smem + 2 * OV_SLAB);
    for (int i = tid_; i < CO; i += NTHR) sBias[i] = p.b3[i];
#pragma unroll 1
    for (int r = 0; r < p.R; ++r) {
        const int panel = panel_of(r);
        if (panel >= p.NP) break;
        int lane = lane_, tid = tid_;
        asm volatile("" : "+v"(lane), "+v"(tid));
        const int l15 = lane & 15, quad = lane >> 4;
    auto issue_slab = [&](int g, int slot) {
        const uint32_t base = lds0 + (uint32_t)(slot * OV_SLAB * 16) + (uint32_t)(wave * 64 * 16);
#pragma unroll
        for (int i = 0; i < OV_SLAB / NTHR; ++i) {
            const int c = i * NTHR + tid;
            const int row = c / CPR, pc = c % CPR;
            const int lc = pc ^ (row & 15);
            const int blk = row >> 4, rho = row & 15;
            const int ch = g * 64 + (blk >> 1) * 32 + (rho >> 2) * 8 + (blk & 1) * 4 + (rho & 3);
            dma16u(p.w3 + (long)ch * CM + lc * 8, __builtin_amdgcn_readfirstlane(base + (uint32_t)(i * NTHR * 16)));
        }
    };
    const int q = quad;
        const int bm0 = panel * BM;
        const long pm = (long)bm0 + wave * 16 + l15;           // this lane's pixel
        const long pmc = pm < (long)p.M ? pm : (long)p.M - 1;
        const bool full = (long)bm0 + wave * 16 + 16 <= (long)p.M;
        stamp(7);
        issue_slab(0, 0);                                      // the first filter slab travels while the partner finishes conv2
        ov_wait_flag(p.flags + panel, p.err, tid);
        stamp(8);
        // this wave's 16 pixels, all of k: lane (l15, quad) = row l15, the chunks 4 ks + quad (write-through data: sc1 loads)
        u32x4 xf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xf[ks] = __builtin_amdgcn_raw_buffer_load_b128(t2rs, (int)((pmc * CM + (ks * 4 + quad) * 8) * 2), 0, 16);
        auto run = [&](auto role) {
            constexpr bool trail = decltype(role)::value;
            u32x4 rr[2];
            auto load_res = [&](int g) {
                const int c0 = (g < GR ? g : GR - 1) * 64 + q * 8;
                rr[0] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)((pmc * CO + c0) * 2), 0, 0);
                rr[1] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)((pmc * CO + c0 + 32) * 2), 0, 0);
            };
            load_res(0);
            f32x4 acc[4];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // slab 0, the T2 fragments, the first residual
            ov_barrier();
            auto gemm = [&](int g) {
                const u32x4 *slab = ov_smem + (g & 1) * OV_SLAB;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                u32x4 wf[2][4];
                auto read_w = [&](int ks, u32x4 (&w)[4]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = slab[(i * 16 + l15) * CPR + ((ks * 4 + q) ^ l15)];
                };
                read_w(0, wf[0]);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 1 < KS) read_w(ks + 1, wf[(ks + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = ov_mfma<F16>(wf[ks & 1][i], xf[ks], acc[i]);
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
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int e = i * 4 + e4;
                        v[e] = acc[i][e4] + bias[e4];
                        const uint32_t rw = rr[e / 8][(e % 8) / 2];
                        v[e] += ov_unpack<F16>((e & 1) ? (rw >> 16) : (rw & 0xffffu));
                        v[e] = fmaxf(v[e], 0.0f);
                    }
                }
                u32x4 yf[2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) yf[k][e] = usot_pack2_lp<F16>(v[k * 8 + 2 * e], v[k * 8 + 2 * e + 1]);
                if (full || pm < (long)p.M) {                  // written through: the M partner reads these rows back for conv1
                    __builtin_amdgcn_raw_buffer_store_b128(yf[0], yrs, (int)((pm * CO + c0) * 2), 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(yf[1], yrs, (int)((pm * CO + c0 + 32) * 2), 0, 16);
                }
            };
            auto wait_vm = [&](int n) {
                if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            };
            // one interval.  Leaders: GEMM(k), epilogue(k), residual of k + 1; trailers: epilogue(k - 1), residual of k, GEMM(k).
            // Slab k + 1 goes to the slot slab k - 1 left: every wave finished reading that one before the barrier that ended interval k - 1.
            auto interval = [&](int k) {
                if (k + 1 < GR) issue_slab(k + 1, (k + 1) & 1);
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
                // slab k + 1 must have landed before the barrier: at most the operations issued after it may be outstanding - this
                // interval's stores (2) and residual loads (2)
                if (!full) wait_vm(0);
                else       wait_vm(stored ? 4 : 0);
                ov_barrier();
            };
#pragma unroll 1
            for (int k = 0; k < GR; ++k) interval(k);
            if constexpr (trail) epilogue(GR - 1);
        };
        if (wave >= NW / 2) run(std::true_type{});
        else                run(std::false_type{});
        ov_publish(p.flags + p.NP + panel, tid);               // the panel's Y rows are in memory
        stamp(9);
    }
}

}  // namespace
#define OV_BUILT 1
#else
#define OV_BUILT 0
constexpr int OV_BM = 128, OV_CM = 256;
#endif

static long long *g_ov_dbg = nullptr;
/* tracing (scripts/conv_pw_ov_probe.py --trace): a device buffer of 32 int64 per workgroup (2 x 8 x ceil(pairs / 8) workgroups) that
 * later launches fill with placement and phase time stamps; NULL switches it off */
extern "C" int usot_conv_pw_ov_trace(void *buf) { g_ov_dbg = (long long *)buf; return USOT_OK; }

extern "C" int usot_conv_pw_ov_supported(int CM, int CO, int CN) { return OV_BUILT && CM == 256 && CO == 1024 && CN == 256; }

/* bytes of zero-initialised scratch a launch over M pixels needs (the T2 hand-off map, the hand-off flags, the error word) */
extern "C" int64_t usot_conv_pw_ov_ws_bytes(int64_t M)
{
    if (M <= 0) return 0;
    const int64_t np = (M + OV_BM - 1) / OV_BM;
    return ((2 * np + 1) * 4 + 255) / 256 * 256 + M * OV_CM * 2;
}

/* The overlapped form of usot_conv_pw_pair_lp for layer3's blocks (CM = 256, CO = 1024, CN = 256; conv2 3 x 3 / stride 1 / pad = dil in
 * 1..4): same arguments + `ws` (usot_conv_pw_ov_ws_bytes(M) bytes, ZERO before the first launch; the launch leaves its flags zero).
 * Returns USOT_EINVAL for geometry it does not take (the caller falls back to usot_conv_pw_pair_lp). */
extern "C" int usot_conv_pw_ov_lp(void *stream, const usot_conv_desc *c2, const usot_pw_pair_desc *d, int dtype, void *ws)
{
#if !OV_BUILT
    (void)stream; (void)c2; (void)d; (void)dtype; (void)ws; (void)g_ov_dbg;
    return USOT_ENOTBUILT;
#else
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    if (!c2 || !d || !ws || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    if (!d->w3p || !d->b3 || !d->res || !d->y || !d->w1 || !d->b1 || !d->t) return USOT_EINVAL;
    if (d->act2 != USOT_ACT_NONE && d->act2 != USOT_ACT_RELU) return USOT_EINVAL;
    if (!usot_conv_pw_ov_supported(d->CM, d->CO, d->CN) || c2->Cout != d->CM || c2->Cin != d->CM) return USOT_EINVAL;
    if (!c2->x || !c2->w || !c2->bias || c2->act != USOT_ACT_RELU || c2->N <= 0 || c2->groups > 1 || c2->ksplit > 1 || c2->res) return USOT_EINVAL;
    if (c2->KH != 3 || c2->KW != 3 || c2->stride != 1 || c2->pad_h != c2->dil_h || c2->pad_w != c2->dil_w || c2->dil_w < 1 || c2->dil_w > 4 ||
        c2->dil_h < 1 || c2->OH != c2->H || c2->OW != c2->W) return USOT_EINVAL;
    const void *ptrs[] = {c2->x, c2->w, c2->bias, d->w3p, d->b3, d->res, d->y, d->w1, d->b1, d->t, ws};
    for (const void *q : ptrs)
        if ((uintptr_t)q % 16) return USOT_EINVAL;
    const long M = (long)c2->N * c2->OH * c2->OW;
    if (M != d->M || M * OV_CO * 2 >= 0x7fffffffL || (long)c2->N * c2->H * c2->W * OV_CM >= 0x7fffffffL) return USOT_EINVAL;   // 32-bit offsets
    static const uint16_t *zero_page_d[USOT_MAX_DEV] = {};
    const uint16_t *&zero_page = zero_page_d[usot_dv];
    static int cus_d[USOT_MAX_DEV] = {};
    int &cus = cus_d[usot_dv];
    if (!zero_page) {
        void *zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(ov_zero16)) != hipSuccess || !zp) return USOT_ELAUNCH;
        zero_page = (const uint16_t *)zp;
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    OvK p = {};
    p.x = (const uint16_t *)c2->x; p.w2 = (const uint16_t *)c2->w; p.zero = zero_page; p.b2 = c2->bias;
    p.w3 = (const uint16_t *)d->w3p; p.res = (const uint16_t *)d->res; p.b3 = d->b3; p.y = (uint16_t *)d->y;
    p.w1 = (const uint16_t *)d->w1; p.b1 = d->b1; p.t = (uint16_t *)d->t; p.act2 = d->act2;
    p.H = c2->H; p.W = c2->W; p.OW = c2->OW; p.pad_h = c2->pad_h; p.dil_h = c2->dil_h; p.dil_w = c2->dil_w;
    p.M = (int)M; p.P = c2->OH * c2->OW;
    p.NP = (int)((M + OV_BM - 1) / OV_BM);
    p.R = (p.NP + cus - 1) / cus;                            // panels per pair: every CU hosts one pair
    const int G = (p.NP + p.R - 1) / p.R;
    p.G8 = (G + 7) / 8 * 8;
    p.flags = (int *)ws;
    p.err = p.flags + 2 * p.NP;
    p.dbg = g_ov_dbg;
    p.t2s = (uint16_t *)((char *)ws + ((2 * (long)p.NP + 1) * 4 + 255) / 256 * 256);
    static bool raised_d[USOT_MAX_DEV][2] = {};
    bool (&raised)[2] = raised_d[usot_dv];
    const void *fn = dtype ? (const void *)conv_pw_ov_kernel<true> : (const void *)conv_pw_ov_kernel<false>;
    if (!raised[dtype]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, OV_LDS) != hipSuccess) return USOT_ELAUNCH;
        raised[dtype] = true;
    }
    if (dtype) hipLaunchKernelGGL(conv_pw_ov_kernel<true>, dim3(2 * p.G8), dim3(OV_NTHR), OV_LDS, (hipStream_t)stream, p);
    else       hipLaunchKernelGGL(conv_pw_ov_kernel<false>, dim3(2 * p.G8), dim3(OV_NTHR), OV_LDS, (hipStream_t)stream, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
#endif
}
