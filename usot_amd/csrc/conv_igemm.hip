// Convolution as implicit GEMM on the gfx950 fp32 matrix cores.
//
// GEMM view (per group):  D[co][m] = sum_k Wp[co][k] * A[m][k]
//   m = (n, oh, ow) output pixel,  k = (kh, kw, ci) with ci innermost (NHWC input),
//   A[m][k] = x[n][oh*s - ph + kh*dh][ow*s - pw + kw*dw][ci]   (0 outside the image).
// The weight tile is the MFMA "A" operand and the activation tile the "B" operand, so a
// lane's four accumulator registers are four CONSECUTIVE output channels of one pixel:
// the epilogue (bias, residual, activation) and the NHWC store are 16-byte accesses.
//
// v_mfma_f32_16x16x4_f32 is exact fp32 (an fmaf chain, one rounding per product) at
// 64 FLOP/clk/SIMD = 157.3 TFLOP/s chip peak: this is the roofline the fp32 path is
// measured against.  Each lane feeds it one float of W and one of A per instruction; a
// lane reads 4 consecutive k of its row with one ds_read_b128 and issues 4 MFMAs from
// it (lane quad q owns k-slot q of every 4x16 step, so a 16-wide k round costs one
// b128 read per 16x16 block row/column).
//
// Tiling: 256 threads = 4 wavefronts; workgroup tile BM(pixels) x BN(channels) x 32(k);
// register-staged double buffering: global loads for tile t+1 are issued before the
// MFMAs of tile t and written to the other LDS buffer after them; one barrier per tile.
// LDS rows are padded to 36 floats (144 B, keeps 16-B alignment, spreads rows over banks).
// Small M (batch 1: M = 961 or 625) is handled by small tiles and optional split-K, not
// by padding up to a big tile — see DESIGN.md "filling 1024 SIMDs at M = 961".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

struct ConvK {
    const float *x, *w, *bias, *res;
    float *y, *ws;
    int N, H, W, Cin, OH, OW, Cout;
    int KH, KW, stride, pad_h, pad_w, dil_h, dil_w;
    int y_cstride, y_coff, res_cstride, res_coff, y_nchw;
    int act, act2, act_split;
    int groups;
    long x_gs, w_gs, b_gs, y_gs, r_gs;
    int ksplit;
    int combine;                        // split-K: 1 = last-arriver combine inside the launch, 0 = second launch
    int M, K, KT, cchunks, MT, NT, P;   // P = OH*OW
    int vec_store;
    int pr;                             // weight-stationary tiles: pixel ranges per (group, 32-channel block)
    const float *wscale;                // split-fp16 tiles (PF = 4): 1 / (filter row scale x activation scale) per output channel
    const float *zero;                  // PF = 6: 16 zero bytes, the source of padding taps for the activation DMA
    int x_split, y_split;               // the input map arrives / the result leaves in the split-fp16 layout (usot_conv_desc)
    int *ovf;                           // split-fp16 tiles: sticky "a finished sum was not finite" word (usot_conv_desc.ovf) or nullptr
};

// Up to four convolutions of DIFFERENT geometry in one launch (same tile shape): the shortcut
// conv beside conv1 of a bottleneck, the three dilated encoders of one input, the two
// prediction heads.  At batch 1 each of these alone leaves most of the chip idle and pays a
// full launch; side by side they share one.  Workgroup ranges are [start[i], start[i+1]).
struct ConvBatch {
    ConvK p[4];
    int n;
    int start[5];
};

__device__ __forceinline__ float apply_act(float v, int a)
{
    switch (a) {
    case USOT_ACT_RELU: return fmaxf(v, 0.0f);
    case USOT_ACT_EXP:  return expf(v);
    case USOT_ACT_CONF: return expf(fminf(fmaxf(v, 0.0f), 4.0f));
    default:            return v;
    }
}

// write-through (sc1) slab accesses through a buffer descriptor: the partial tiles of a split-K launch bypass the
// issuing CU's L1 and are not left dirty in its XCD's L2, so the hand-off needs NO release fence (buffer_wbl2 writes
// back the whole L2 slice: 1.7-6.5 us) and the reader no acquire (cdna_hip_programming.md Guideline 16, recipe R1)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const ConvK &p)
{
    const long bytes = ((long)p.ksplit * p.groups * p.M * p.Cout) * 4;
    return __builtin_amdgcn_make_buffer_rsrc((void *)p.ws, 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}
__device__ __forceinline__ void ws_store4(const ConvK &p, long elem, f32x4 v)
{
    if (p.combine) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ws_rsrc(p), (int)(elem * 4), 0, 16);
    else           *(f32x4 *)(p.ws + elem) = v;
}
__device__ __forceinline__ void ws_store1(const ConvK &p, long elem, float v)
{
    if (p.combine) __hip_atomic_store(p.ws + elem, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else           p.ws[elem] = v;
}

// In-launch split-K combine.  Every k-slice workgroup has stored its partial tile to the workspace slab
// [ks][group][M][Cout]; the LAST slice to arrive (one ticket per tile) sums the slabs in slice order — the
// same order, hence the same bits, whichever workgroup does it — and applies bias / residual / activation.
// Replaces the separate reduction launch (4.8 us + a kernel boundary per split layer at batch 1).
// Protocol (cdna_hip_programming.md, Guideline 16 recipe R1 / "in-launch split-K reduction"): write-through (sc1)
// slab stores -> every wave drains vmcnt -> barrier -> ONE lane: relaxed agent-scope fetch_add -> last arriver:
// barrier -> sc1 slab loads.  No fences (the release form, buffer_wbl2, measured 14 us per frame SLOWER than the
// separate reduction launch it replaces).  Placement-independent.  The ticket words live
// behind the slabs, are zero before the first launch (the engine zero-fills the workspace) and are reset by
// the last arriver, so every launch — and every graph replay — finds them zero.
template <int BM, int BN>
__device__ __forceinline__ void splitk_combine(const ConvK &p, int g, int t0, int tiles, int bm0, int bn0, int tid, int *flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        int *cnt = (int *)(p.ws + (long)p.ksplit * p.groups * p.M * p.Cout) + (g * tiles + t0);
        const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == p.ksplit - 1;
        if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    const float *__restrict__ bg = p.bias ? p.bias + (long)g * p.b_gs : nullptr;
    const float *__restrict__ rg = p.res ? p.res + (long)g * p.r_gs : nullptr;
    float *__restrict__ yg = p.y + (long)g * p.y_gs;
    const long slab = (long)p.groups * p.M * p.Cout;
    const float *ws0 = p.ws + (long)g * p.M * p.Cout;
    const __amdgpu_buffer_rsrc_t rs = ws_rsrc(p);
    if (p.vec_store && (p.Cout & 3) == 0) {
        for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
            const int row = idx / (BN / 4), c4 = (idx - row * (BN / 4)) * 4;
            const int m = bm0 + row, co = bn0 + c4;
            if (m >= p.M || co >= p.Cout) continue;
            const long el = (long)g * p.M * p.Cout + (long)m * p.Cout + co;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < p.ksplit; ++ks)
                v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((el + ks * slab) * 4), 0, 16));
            if (bg) v += *(const f32x4 *)(bg + co);
            if (rg) v += *(const f32x4 *)(rg + (long)m * p.res_cstride + p.res_coff + co);
            const int a = co < p.act_split ? p.act : p.act2;
            if (a != USOT_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a);
            }
            *(f32x4 *)(yg + (long)m * p.y_cstride + p.y_coff + co) = v;
        }
        return;
    }
    for (int idx = tid; idx < BM * BN; idx += 256) {
        const int row = idx / BN, cc = idx - row * BN;
        const int m = bm0 + row, c = bn0 + cc;
        if (m >= p.M || c >= p.Cout) continue;
        float sum = 0.f;
        for (int ks = 0; ks < p.ksplit; ++ks) sum += __hip_atomic_load(ws0 + ks * slab + (long)m * p.Cout + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bg) sum += bg[c];
        if (rg) sum += rg[(long)m * p.res_cstride + p.res_coff + c];
        sum = apply_act(sum, c < p.act_split ? p.act : p.act2);
        if (p.y_nchw) {
            const int n = m / p.P, pix = m - n * p.P;
            yg[((long)n * p.Cout + c) * p.P + pix] = sum;
        } else {
            yg[(long)m * p.y_cstride + p.y_coff + c] = sum;
        }
    }
}

// XCD-aware, bijective block remap: the dispatcher places block b on XCD b % 8; give each
// XCD a contiguous chunk of the tile space so the M-tiles that share a weight strip (and
// the neighbouring pixels' input rows) meet in one L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int total)
{
    const int q = total >> 3, r = total & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

constexpr int LDK = 36;   // padded k-extent of an LDS row (floats)
__device__ __attribute__((aligned(16))) uint32_t g_zero16_f32[4] = {0u, 0u, 0u, 0u};     // source of padding taps (activation DMA, PF = 6)

// Blocked accumulation.  v_mfma_f32_16x16x4_f32 adds its four products to C one after the other, so a plain k-loop is ONE
// sequential float32 chain of K additions per output: on a K = 4608 reduction its rounding error is 4x (rms) that of a
// CPU convolution of the same data, and with filters that pass the DC level of post-ReLU maps the frame's logits end up
// 2.8x further from a float64 evaluation than the reference's own float32 arithmetic (tests/test_gpu_model.py, 'dc' family).
// Every kernel of this file therefore sums a k-tile (32 / 64 products) into a fresh accumulator — the first MFMA of a block
// takes C = 0, an inline constant: nothing to clear — and adds finished blocks to a running total (common.h: BlockTotal;
// plain float32 by default, float64 / Kahan forms measured there), eight VALU adds per k-tile placed right after the
// k-loop's barrier where the block's MFMAs have long retired, issued in the shadow of the next block's MFMAs.  Chains:
// 64 + K/64 instead of K.  Emulated on the layer3.0 shortcut conv: rms error of the output 1.76e-6 (single
// chain) -> 3.6e-7 (float32 total; torch-CPU: 3.9e-7); measured on the whole frame in tests/golden/f64_gate.py.
// one 16-k round = 4 MFMAs per 16x16 block from the fragments wf / xf; first: this round starts a block
template <int TN, int TM, bool DUAL = true>
__device__ __forceinline__ void blocked_mma(f32x4 (&acc)[TN][TM], f32x4 &acc2, const f32x4 (&wf)[TN], const f32x4 (&xf)[TM], bool first)
{
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if constexpr (TM * TN == 1 && DUAL) {
        // a wave with a single 16x16 block alternates two accumulators (40-cycle dependent latency vs 32-cycle issue)
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0][0], xf[0][0], first ? zero : acc[0][0], 0, 0, 0);
        acc2      = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0][1], xf[0][1], first ? zero : acc2, 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0][2], xf[0][2], acc[0][0], 0, 0, 0);
        acc2      = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0][3], xf[0][3], acc2, 0, 0, 0);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][c], xf[j][c], (first && c == 0) ? zero : acc[i][j], 0, 0, 0);
    }
}

template <int BM, int BN, int WM, int WN, int DBG = 0>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvBatch bt)
{
    int pi = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < bt.n && (int)blockIdx.x >= bt.start[q]) pi = q;
    const ConvK &p = bt.p[pi];
    const int bid0 = (int)blockIdx.x - bt.start[pi];
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    static_assert(TM >= 1 && TN >= 1, "wave tile at least 16x16");
    constexpr int XI = (BM + 31) / 32, WI = (BN + 31) / 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sX = smem;                   // [2][BM][LDK]
    float *sW = smem + 2 * BM * LDK;    // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int tiles = p.MT * p.NT;
    const int total = tiles * p.groups * p.ksplit;
    const int b = xcd_remap(bid0, total);
    const int z = b / tiles, t = b - z * tiles;
    const int g = z / p.ksplit, ks = z - g * p.ksplit;
    const int bn0 = (t / p.MT) * BN, bm0 = (t % p.MT) * BM;
    const int kt0 = (int)((long)p.KT * ks / p.ksplit);
    const int kt1 = (int)((long)p.KT * (ks + 1) / p.ksplit);

    const float *__restrict__ xg = p.x + (long)g * p.x_gs;
    const float *__restrict__ wg = p.w + (long)g * p.w_gs;

    // ---- loader role: thread (lr, kc) moves 16-byte chunk kc of rows lr, lr+32, ...
    const int lr = tid >> 3, kc = tid & 7;
    int x_ih0[XI], x_iw0[XI];
    long x_nb[XI];
    bool x_ok[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = lr + 32 * i;
        const int m = bm0 + row;
        x_ok[i] = (row < BM) && (m < p.M);
        const int mm = x_ok[i] ? m : 0;
        const int n = mm / p.P, pix = mm - n * p.P;
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        x_ih0[i] = oh * p.stride - p.pad_h;
        x_iw0[i] = ow * p.stride - p.pad_w;
        x_nb[i] = (long)n * p.H * p.W * p.Cin + kc * 4;
    }
    long w_off[WI];
    bool w_ok[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = lr + 32 * i;
        const int co = bn0 + row;
        w_ok[i] = (row < BN) && (co < p.Cout);
        w_off[i] = (long)(w_ok[i] ? co : 0) * p.K + kc * 4;
    }

    f32x4 xr[XI], wr[WI];
    auto load_tile = [&](int kt) {
        const int tap = kt / p.cchunks, c0 = (kt - tap * p.cchunks) * 32;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const int dh = kh * p.dil_h, dw = kw * p.dil_w;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int ih = x_ih0[i] + dh, iw = x_iw0[i] + dw;
            const bool ok = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4 *)(xg + x_nb[i] + ((long)ih * p.W + iw) * p.Cin + c0);
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (w_ok[i]) v = *(const f32x4 *)(wg + w_off[i] + (long)kt * 32);
            wr[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XI; ++i)
            if (lr + 32 * i < BM)
                *(f32x4 *)(sX + (buf * BM + lr + 32 * i) * LDK + kc * 4) = xr[i];
#pragma unroll
        for (int i = 0; i < WI; ++i)
            if (lr + 32 * i < BN)
                *(f32x4 *)(sW + (buf * BN + lr + 32 * i) * LDK + kc * 4) = wr[i];
    };

    // ---- MFMA role
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l15 = lane & 15, quad = lane >> 4;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    BlockTotal tot[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) tot[i][j].clear();
    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const bool more = kt + 1 < kt1;
        if (more && DBG < 1) load_tile(kt + 1);
        const float *cX = sX + (cur * BM + wm * TM * 16 + l15) * LDK + quad * 4;
        const float *cW = sW + (cur * BN + wn * TN * 16 + l15) * LDK + quad * 4;
        if (((kt - kt0) & 1) == 0) {                  // a block = two 32-deep k-tiles (blocked_mma above)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) tot[i][j].add(acc[i][j]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x4 wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = DBG < 2 ? *(const f32x4 *)(cW + i * 16 * LDK + r * 16) : f32x4{1.f + kt, 2.f, 3.f, 4.f + i};
#pragma unroll
            for (int j = 0; j < TM; ++j) xf[j] = DBG < 2 ? *(const f32x4 *)(cX + j * 16 * LDK + r * 16) : f32x4{1.f, 2.f + kt, 3.f + j, 4.f};
            f32x4 unused = {0.f, 0.f, 0.f, 0.f};
            blocked_mma<TN, TM, false>(acc, unused, wf, xf, r == 0 && ((kt - kt0) & 1) == 0);
        }
        if (more && DBG < 1) store_tile(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) { tot[i][j].add(acc[i][j]); acc[i][j] = tot[i][j].get(); }

    // ---- epilogue: lane holds channels co..co+3 (quad*4 + reg) of pixel m (l15)
    if (p.ksplit > 1) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = bm0 + (wm * TM + j) * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int co = bn0 + (wn * TN + i) * 16 + quad * 4;
                const long el = ((long)(ks * p.groups + g) * p.M + m) * p.Cout + co;
                if (co + 3 < p.Cout && (p.Cout & 3) == 0) {
                    ws_store4(p, el, acc[i][j]);
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Cout) ws_store1(p, el + e, acc[i][j][e]);
                }
            }
        }
        if (p.combine) splitk_combine<BM, BN>(p, g, t, tiles, bm0, bn0, tid, (int *)smem);
        return;
    }
    const float *__restrict__ bg = p.bias ? p.bias + (long)g * p.b_gs : nullptr;
    const float *__restrict__ rg = p.res ? p.res + (long)g * p.r_gs : nullptr;
    float *__restrict__ yg = p.y + (long)g * p.y_gs;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = bm0 + (wm * TM + j) * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int co = bn0 + (wn * TN + i) * 16 + quad * 4;
            if (co >= p.Cout) continue;
            f32x4 v = acc[i][j];
            if (p.vec_store && co + 3 < p.Cout) {
                if (bg) v += *(const f32x4 *)(bg + co);
                if (rg) v += *(const f32x4 *)(rg + (long)m * p.res_cstride + p.res_coff + co);
                const int a = co < p.act_split ? p.act : p.act2;
                if (a != USOT_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a);
                }
                *(f32x4 *)(yg + (long)m * p.y_cstride + p.y_coff + co) = v;
            } else {
                const int n = m / p.P, pix = m - n * p.P;
                for (int e = 0; e < 4; ++e) {
                    const int c = co + e;
                    if (c >= p.Cout) break;
                    float s = v[e];
                    if (bg) s += bg[c];
                    if (rg) s += rg[(long)m * p.res_cstride + p.res_coff + c];
                    s = apply_act(s, c < p.act_split ? p.act : p.act2);
                    if (p.y_nchw) yg[((long)n * p.Cout + c) * p.P + pix] = s;
                    else          yg[(long)m * p.y_cstride + p.y_coff + c] = s;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// v2 main loop: three LDS stages, the barrier in the MIDDLE of a k-tile, fragments
// prefetched one round ahead (also across the k-tile boundary).
//
// At batch 1 most layers run one wavefront per SIMD, so nothing hides a wave's own waits:
// every cycle it spends on ds_write -> barrier -> ds_read -> first MFMA is a cycle the matrix
// pipe idles (measured on v1, 64x64 tile, layer3.0 shortcut: MFMA-only 130 TFLOP/s, with
// staging 58).  Here, per k-tile t:
//     F1 <- fragments(t, round 1)            issued before round 0's MFMAs
//     MFMA round 0 (F0)
//     G (tile t+1, loaded during tile t-1) -> LDS[(t+1)%3];  issue global loads of t+2 -> G
//     barrier                                 (tile t+1 visible; nobody still reads (t+1)%3,
//                                              it held tile t-2)
//     F0 <- fragments(t+1, round 0)          hidden behind round 1's MFMAs
//     MFMA round 1 (F1)
// so the global-load latency has a whole k-tile to land, the LDS write is off the critical
// path, and no MFMA ever waits for a just-issued ds_read.
template <int BM, int BN, int WM, int WN, int BK, int KSW = 1>
__global__ __launch_bounds__(256 * KSW) void conv_igemm_f32_v2(const ConvBatch bt)
{
    int pi = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < bt.n && (int)blockIdx.x >= bt.start[q]) pi = q;
    const ConvK &p = bt.p[pi];
    const int bid0 = (int)blockIdx.x - bt.start[pi];
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    static_assert(BK == 32 || BK == 64, "k-tile of 32 or 64");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int LD = BK + 4;
    constexpr int CPR = BK / 4;                 // 16-byte chunks per row
    constexpr int RPP = 256 / CPR;              // rows covered per pass of the 256 loaders
    constexpr int XI = (BM + RPP - 1) / RPP, WI = (BN + RPP - 1) / RPP;
    constexpr int NR = BK / 16;                 // MFMA rounds per k-tile
    constexpr int STAGE = (BM + BN) * LD;

    // KSW > 1: the workgroup has KSW groups of 4 wavefronts that share ONE output tile and
    // split its k-tiles round-robin (group g takes kt0+g, kt0+g+KSW, ...), each with its own
    // LDS stages; partial accumulators meet in LDS at the end.  Two (or four) waves per SIMD
    // from the same tile hide each other's staging without a cross-workgroup reduction, a
    // workspace round trip or a second launch (what inter-workgroup split-K costs).
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int grp = KSW > 1 ? (int)(threadIdx.x >> 8) : 0;
    float *smem = smem_all + grp * 3 * STAGE;

    const int tid = threadIdx.x & 255;
    const int tiles = p.MT * p.NT;
    const int total = tiles * p.groups * p.ksplit;
    const int b = xcd_remap(bid0, total);
    const int z = b / tiles, t0 = b - z * tiles;
    const int g = z / p.ksplit, ks = z - g * p.ksplit;
    const int bn0 = (t0 / p.MT) * BN, bm0 = (t0 % p.MT) * BM;
    const int KT = p.K / BK;                    // k-tiles of this instantiation
    const int cch = p.Cin / BK;
    const int kt0 = (int)((long)KT * ks / p.ksplit);
    const int kt1 = (int)((long)KT * (ks + 1) / p.ksplit);

    const float *__restrict__ xg = p.x + (long)g * p.x_gs;
    const float *__restrict__ wg = p.w + (long)g * p.w_gs;

    const int lr = tid / CPR, kc = tid % CPR;
    int x_ih0[XI], x_iw0[XI];
    long x_nb[XI];
    bool x_ok[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = lr + RPP * i;
        const int m = bm0 + row;
        x_ok[i] = (row < BM) && (m < p.M);
        const int mm = x_ok[i] ? m : 0;
        const int n = mm / p.P, pix = mm - n * p.P;
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        x_ih0[i] = oh * p.stride - p.pad_h;
        x_iw0[i] = ow * p.stride - p.pad_w;
        x_nb[i] = (long)n * p.H * p.W * p.Cin + kc * 4;
    }
    long w_off[WI];
    bool w_ok[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = lr + RPP * i;
        const int co = bn0 + row;
        w_ok[i] = (row < BN) && (co < p.Cout);
        w_off[i] = (long)(w_ok[i] ? co : 0) * p.K + kc * 4;
    }

    // Loader state is incremental: a row's source pointer is recomputed only when the filter
    // tap changes (every Cin/BK k-tiles, never for a 1x1 conv); inside a tap the next k-tile
    // is just +BK floats.  Keeps the per-k-tile VALU work (exposed at one wave per SIMD) small.
    f32x4 xr[XI], wr[WI];
    const float *xp[XI];
    bool xin[XI];
    const float *wp[WI];
    int cur_tap = (kt0 + grp) / cch, cur_cc = (kt0 + grp) - cur_tap * cch;
    auto set_tap = [&](int tap) {
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const int dh = kh * p.dil_h, dw = kw * p.dil_w;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int ih = x_ih0[i] + dh, iw = x_iw0[i] + dw;
            xin[i] = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            xp[i] = xg + x_nb[i] + ((long)ih * p.W + iw) * p.Cin;
        }
    };
    set_tap(cur_tap);
#pragma unroll
    for (int i = 0; i < WI; ++i) wp[i] = wg + w_off[i] + (long)(kt0 + grp) * BK;
    auto load_tile = [&](bool advance) {        // loads the NEXT k-tile in sequence
        const int c0 = cur_cc * BK;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (xin[i]) v = *(const f32x4 *)(xp[i] + c0);
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            // rows >= Cout read row 0: they only feed accumulators that are never stored
            wr[i] = *(const f32x4 *)wp[i];
            wp[i] += advance ? KSW * BK : 0;
        }
        if (advance) {
            cur_cc += KSW;
            if (cur_cc >= cch) {
                do { cur_cc -= cch; ++cur_tap; } while (cur_cc >= cch);
                set_tap(cur_tap);
            }
        }
    };
    auto store_tile = [&](int st) {
        float *sX = smem + st * STAGE, *sW = sX + BM * LD;
#pragma unroll
        for (int i = 0; i < XI; ++i)
            if (BM % RPP == 0 || lr + RPP * i < BM) *(f32x4 *)(sX + (lr + RPP * i) * LD + kc * 4) = xr[i];
#pragma unroll
        for (int i = 0; i < WI; ++i)
            if (BN % RPP == 0 || lr + RPP * i < BN) *(f32x4 *)(sW + (lr + RPP * i) * LD + kc * 4) = wr[i];
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l15 = lane & 15, quad = lane >> 4;
    const int fx_off = (wm * TM * 16 + l15) * LD + quad * 4;
    const int fw_off = BM * LD + (wn * TN * 16 + l15) * LD + quad * 4;
    f32x4 fw[2][TN], fx[2][TM];
    auto read_frags = [&](int st, int r, int slot) {
        const float *base = smem + st * STAGE + r * 16;
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[slot][i] = *(const f32x4 *)(base + fw_off + i * 16 * LD);
#pragma unroll
        for (int j = 0; j < TM; ++j) fx[slot][j] = *(const f32x4 *)(base + fx_off + j * 16 * LD);
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A wave with a single 16x16 block would chain every MFMA on one accumulator (40-cycle
    // dependent latency vs 32-cycle issue: an 80 % cap); it alternates two accumulators instead
    // (even / odd k-slots) and adds them at the end.
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
    // Blocked accumulation (blocked_mma above).  The block boundary sits at the mid-tile barrier: right after it the MFMAs
    // issued before it have retired, the block is added to `tot` and the tile's last round starts the next block with C = 0.
    BlockTotal tot[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) tot[i][j].clear();
    auto flush = [&]() {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) tot[i][j].add(acc[i][j]);
        if constexpr (TM * TN == 1) tot[0][0].add(acc2);
    };
    auto mma = [&](int slot, bool first) { blocked_mma<TN, TM>(acc, acc2, fw[slot], fx[slot], first); };

    const int nt_all = kt1 - kt0;
    const int nt = (nt_all - grp + KSW - 1) / KSW;      // k-tiles of this wave group
    const int nt_max = (nt_all + KSW - 1) / KSW;        // barriers must match across groups
    if (KSW > 1 && nt <= 0) __syncthreads();
    if (nt > 0) {
        load_tile(nt > 1);
        store_tile(0);
        if (nt > 1) load_tile(nt > 2);
        __syncthreads();
        read_frags(0, 0, 0);
    }
    int st = 0;                                   // stage holding tile t
    for (int t = 0; t < nt_max; ++t) {
        if (KSW > 1 && t >= nt) {                 // this group is out of k-tiles: keep the barrier count
            __syncthreads();
            continue;
        }
        const int st1 = st == 2 ? 0 : st + 1;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r + 1 < NR) read_frags(st, r + 1, (r + 1) & 1);
            if (r == NR - 1) {                     // mid-tile hand-over sits before the LAST round
                if (t + 1 < nt) store_tile(st1);
                if (t + 2 < nt) load_tile(t + 3 < nt);
                __syncthreads();                  // lowers to lgkmcnt(0) + s_barrier (no vmcnt drain)
                if (t + 1 < nt) read_frags(st1, 0, 0);
                flush();
            }
            mma(r & 1, r == NR - 1);
        }
        st = st1;
    }
    flush();
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = tot[i][j].get();

    if (KSW > 1) {
        // meet in LDS: thread (lane, wave) of every group holds the same (channel, pixel) slots
        __syncthreads();
        f32x4 *red = (f32x4 *)smem_all;
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) red[(((grp - 1) * TN + i) * TM + j) * 256 + tid] = acc[i][j];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g2 = 1; g2 < KSW; ++g2)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] += red[(((g2 - 1) * TN + i) * TM + j) * 256 + tid];
    }
    // ---- epilogue (same as v1)
    if (p.ksplit > 1) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = bm0 + (wm * TM + j) * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int co = bn0 + (wn * TN + i) * 16 + quad * 4;
                const long el = ((long)(ks * p.groups + g) * p.M + m) * p.Cout + co;
                if (co + 3 < p.Cout && (p.Cout & 3) == 0) {
                    ws_store4(p, el, acc[i][j]);
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Cout) ws_store1(p, el + e, acc[i][j][e]);
                }
            }
        }
        if (p.combine) splitk_combine<BM, BN>(p, g, t0, tiles, bm0, bn0, tid, (int *)smem_all);
        return;
    }
    const float *__restrict__ bg = p.bias ? p.bias + (long)g * p.b_gs : nullptr;
    const float *__restrict__ rg = p.res ? p.res + (long)g * p.r_gs : nullptr;
    float *__restrict__ yg = p.y + (long)g * p.y_gs;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = bm0 + (wm * TM + j) * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int co = bn0 + (wn * TN + i) * 16 + quad * 4;
            if (co >= p.Cout) continue;
            f32x4 v = acc[i][j];
            if (p.vec_store && co + 3 < p.Cout) {
                if (bg) v += *(const f32x4 *)(bg + co);
                if (rg) v += *(const f32x4 *)(rg + (long)m * p.res_cstride + p.res_coff + co);
                const int a = co < p.act_split ? p.act : p.act2;
                if (a != USOT_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a);
                }
                *(f32x4 *)(yg + (long)m * p.y_cstride + p.y_coff + co) = v;
            } else {
                const int n = m / p.P, pix = m - n * p.P;
                for (int e = 0; e < 4; ++e) {
                    const int c = co + e;
                    if (c >= p.Cout) break;
                    float s = v[e];
                    if (bg) s += bg[c];
                    if (rg) s += rg[(long)m * p.res_cstride + p.res_coff + c];
                    s = apply_act(s, c < p.act_split ? p.act : p.act2);
                    if (p.y_nchw) yg[((long)n * p.Cout + c) * p.P + pix] = s;
                    else          yg[(long)m * p.y_cstride + p.y_coff + c] = s;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// v3: producer / consumer wavefronts.  512 threads: waves 0-3 only feed the matrix pipe
// (ds_read fragments -> MFMA), waves 4-7 only move data (global -> registers -> LDS).
// Each SIMD hosts one of each, so the loader's address math, VMEM issue and LDS writes run in
// the shadow of the other wave's MFMAs instead of inside the MFMA wave's own instruction
// stream — the overlap a one-wave-per-SIMD launch (batch 1) cannot get from occupancy.
// Three LDS stages, one barrier per k-tile; the producers stay two tiles ahead, so at tile t a
// consumer may read tile t and prefetch the first fragments of tile t+1 without waiting:
//     producers, tile t : G (tile t+2) -> stage (t+2)%3 ; issue loads of tile t+3 ; barrier
//     consumers, tile t : rounds of {prefetch next fragments ; MFMAs} ;             barrier
// stage (t+2)%3 held tile t-1, which every consumer finished before barrier(t-1).
// D = k-tiles of global loads a producer keeps in flight in registers: with D = 1 the loads of
// tile t+3 are issued only after those of tile t+2 have landed, so a k-step can never be shorter
// than one L2/HBM round trip (measured: 32x32x64 steps took ~600 ns with the MFMAs removed);
// D = 2/3 rotate that many register buffers so a step only waits for loads issued D steps ago.
// NPW = producer wavefronts (4 or 8): the producers' k-step (four ds_write_b128 + lgkmcnt(0) ~500 cycles, address
// VALU + load issue ~450, measured with s_memtime) is what the consumers wait for; eight producers halve the
// per-wave share.  (Fetching ALL fragments of k-tile t+1 during the MFMAs of tile t, instead of one 16-k round
// ahead, measured 20 % slower and was dropped.)
// (Fetching ALL fragments of k-tile t+1 in front of tile t's MFMAs - pinned with sched_barrier, hipcc otherwise
// sinks the reads to the end of the step - shortens the consumer's MFMA phase 828 -> 712 cycles but the burst of
// ds_read_b128 doubles the producers' ds_write time; k-step 1064 -> 1196.  Not kept.)
// (USOT_V3_SWZ builds: second launch bound = waves per SIMD the register allocation must leave room for)
// PF = rounds (16 k each) a consumer's fragment reads run ahead of their MFMAs: 1 = two fragment slots; 2 = one slot per round
// of the k-tile (BK = 64), reads issued two rounds = 16 MFMAs ahead and pinned there (sched_barrier)
// PF = 4 (round 5): SPLIT-fp16 arithmetic.  Every fp32 operand is carried as hi + lo, two fp16 numbers (22 significant bits:
// hi = fp16(s v), lo = fp16(s v - hi), s a power of two that puts the tensor near the top of fp16's range), and a product
// w x is formed as w_lo x_hi + w_hi x_lo + w_hi x_hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation - three 16-cycle MFMAs
// per 32 k where the fp32 MFMA needs eight 32-cycle ones (5.3 x less matrix-pipe time), the dropped w_lo x_lo term is 2^-22 of
// the product.  Filters arrive pre-split from the host (engine.py: PackedConv.w_split16 - per output row a power-of-two scale,
// each k-tile of 64 stored as 64 hi halves + 64 lo halves = the 256 bytes of the fp32 row segment, so the producers move them
// unchanged); activations are split by the PRODUCER waves as they stage a tile (x 8, two ds_write_b64 per four values: VALU
// of a sibling wave costs the MFMA waves nothing).  An LDS row is [64 hi | 64 lo]: the consumers' four 16-byte fragment reads
// per row and k-tile are the fp32 kernel's (rounds 0 / 1 = hi of k 0-31 / 32-63, rounds 2 / 3 = lo).  The finished sums are
// multiplied by wscale[co] = 1 / (row scale x 8) (exact: powers of two) before anything else sees them.  Range contract:
// |activation| < 8 188 (fp16 overflow above that shows as inf / nan in the output, never silently); accuracy: within the
// 1e-4 bar of the fp32 path and as close to float64 as it (tests/test_gpu_model.py, tests/golden/f64_gate.py).
template <int BM, int BN, int WM, int WN, int BK, int D = 1, int NPW = 4, int PF = 1>
#ifdef USOT_V3_SWZ
__global__ __launch_bounds__(64 * WM * WN + 64 * NPW, (BM * BN <= 32 * 64) ? (4 + NPW) / 2 : 1) void conv_igemm_f32_v3(const ConvBatch bt)
#else
__global__ __launch_bounds__(PF == 6 ? 64 * WM * WN + 768 : 64 * WM * WN + 64 * NPW + (PF == 5 ? 256 : 0)) void conv_igemm_f32_v3(const ConvBatch bt)
#endif
{
    int pi = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < bt.n && (int)blockIdx.x >= bt.start[q]) pi = q;
    const ConvK &p = bt.p[pi];
    const int bid0 = (int)blockIdx.x - bt.start[pi];
    // WM x WN consumer wavefronts: 4 (one per SIMD) or 8 (two per SIMD, round 5: the same tile in smaller wave tiles, nothing to
    // exchange).  Probe (scripts/probes/coissue_probe.hip, inline-asm reads, independent accumulators): a SIBLING wave's VALU / SALU /
    // LDS / VMEM instructions do not delay a wave's back-to-back MFMAs at all (32.0 cycles each); the wave's OWN ds_read_b128, issued
    // a round ahead, cost ~6 cycles each (292 / 286 / 298 / 310 cycles per round of 8 MFMAs with 1 / 2 / 3 / 4 reads); ds_write_b128 of
    // the four sibling waves between a read and its use add +30 cycles per round at 3 per wave (12 per round), +67 at 6 per wave
    // (24 per round) - v3 issues 24 per K-STEP, 6 per round: ~ +15; sibling LDS-DMA pieces (4-8 per round) add nothing.  So the probe
    // accounts for 1 024 + 4 x (42 + 15) = 1 250 of the traced 1 396-cycle MFMA phase; no single mechanism measured here explains the rest.
    // Measured: tower level 29.5 -> 28.2 us, Conf_Fusion's conv 113 -> 114, 64 x 64 tile 49.5 -> 46: not routed.
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 consumer wavefronts");
    constexpr int CT = 64 * WM * WN;              // consumer threads; the producers follow
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr bool H16 = PF == 4 || PF == 5 || PF == 6;      // split-fp16 arithmetic (above)
    static_assert(!H16 || BK == 64, "split-fp16 rows are 64 hi + 64 lo halves");
    // PF = 5: split-fp16 with the FILTER tile moved by LDS-DMA.  The pre-split filter rows are copied to LDS unchanged, and the
    // producers' ds_write_b128 of them were two thirds of the 24 KB a k-step pushes through the VGPR -> LDS store path (~79 B/clk
    // per CU: 310 of the traced 560-cycle store phase the consumers end up waiting for).  Four extra wavefronts (one per SIMD) do
    // nothing but issue global_load_lds_dwordx4 for the filter rows of k-tile t + 4 into a ring of FIVE filter stages (the
    // activation tile keeps its three register-staged stages and its producers, which now move only the 8 KB that need splitting),
    // wait with a counted s_waitcnt vmcnt until the pieces of tile t + 2 have landed, and join the k-step's barrier.  A stage is
    // the same padded row image as before (17 chunks of 16 bytes per row: linear chunk c = 17 row + chunk, the 17th a pad that
    // receives a dummy load), so the consumers' fragment addresses do not change; 1 KiB pieces are c = 64 piece + lane.
    // PF = 6: BOTH operands by LDS-DMA.  The input map arrives ALREADY split (usot_conv_desc.x_split: per pixel and 64-channel block the
    // 64 hi halves then the 64 lo halves of 8 x value - the 256 bytes of the fp32 block, written by the producing kernel's epilogue,
    // y_split), so an activation row of a k-tile is as ready-made as a filter row: the four DMA wavefronts fetch both tiles (a padding
    // tap fetches 16 zero bytes), five stages each, and the producer wavefronts with their loads, conversions and ds_writes are gone.
    constexpr bool XDMA = PF == 6;
    constexpr bool WDMA = PF == 5 || PF == 6;
    constexpr int NDW = XDMA ? 12 : (WDMA ? 4 : 0);  // DMA wavefronts (a piece costs its wave ~200 cycles of issue - traced: 26 pieces per k-tile over twelve waves)
    constexpr int NSW = WDMA ? ((D >= 4 || PF == 6) ? 6 : 5) : 3;   // filter stages (six on the all-DMA tiles and those whose activation producers run four k-tiles deep)
    constexpr int LA = NSW - 1;                   // the DMA runs LA k-tiles ahead
    // LDS rows.  BK = 64: a row is 256 B = one bank row, UNPADDED, 16-byte chunk c of tile row `row` stored at chunk
    // c ^ (row & 15).  ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md
    // LDS table): with the former 4-float row pad, lane (l15, quad) hit 16-byte slot (l15 + quad) mod 16 and every group had
    // two lanes on one slot (lanes 12 / 27, 4 / 19, ...): 8 LDS cycles per fragment read instead of 4.  With the XOR the
    // slots of a group are (4 r) ^ quad ^ l15: {0-3,12-15} ^ 0 and {4-11} ^ 1 - disjoint in every group; the producers' 8-lane
    // ds_write_b128 groups cover 8 consecutive chunks of one row, still 8 distinct slots.  BK = 32 keeps the padded rows.
#ifdef USOT_V3_SWZ        // measured, round 5: conflict-free (SQ_LDS_BANK_CONFLICT 0) and SLOWER - frame graph 871 vs 849 us on the same box
    constexpr bool SWZ = BK == 64 && !H16;   // (one more v_xor per fragment read, 78 -> 80 VGPRs + a spill on the 32 x 64 tile); kept buildable
#else
    constexpr bool SWZ = false;
#endif
    constexpr int LD = SWZ ? BK : BK + 4;
    constexpr int CPR = BK / 4;
    constexpr int NPT = 64 * NPW;                 // producer threads
    constexpr int RPP = NPT / CPR;
    constexpr int XI = (BM + RPP - 1) / RPP, WI = (BN + RPP - 1) / RPP;
    constexpr int NR = BK / 16;
    constexpr int STAGE = (BM + BN) * LD;
    // LDS: without the filter DMA three stages of [activation rows | filter rows]; with it three activation stages, then NSW filter stages
    constexpr int NSX = XDMA ? NSW : 3;                        // activation stages
    constexpr int XSTRIDE = WDMA ? BM * LD : STAGE;            // floats between activation stages
    constexpr int WBASE = WDMA ? NSX * BM * LD : BM * LD;      // first filter stage
    constexpr int WSTRIDE = WDMA ? BN * LD : STAGE;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool producer = threadIdx.x >= CT;
    const int tid = producer ? (int)threadIdx.x - CT : (int)threadIdx.x;
    constexpr int NPTL = XDMA ? 0 : NPT;          // producer threads actually launched (none when the activations come by DMA)
    const int tiles = p.MT * p.NT;
    const int total = tiles * p.groups * p.ksplit;
    const int b = xcd_remap(bid0, total);
    const int z = b / tiles, t0 = b - z * tiles;
    const int g = z / p.ksplit, ks = z - g * p.ksplit;
    const int bn0 = (t0 / p.MT) * BN, bm0 = (t0 % p.MT) * BM;
    const int KT = p.K / BK;
    const int cch = p.Cin / BK;
    const int kt0 = (int)((long)KT * ks / p.ksplit);
    const int kt1 = (int)((long)KT * (ks + 1) / p.ksplit);
    const int nt = kt1 - kt0;

    if constexpr (WDMA) {
        if ((int)threadIdx.x >= CT + NPTL) {
            // ---------------- DMA wavefronts: the filter tile (PF = 5, 6) and, with PF = 6, the activation tile too
            const int dt = (int)threadIdx.x - CT - NPTL;
            const int dwv = __builtin_amdgcn_readfirstlane(dt >> 6), lane = dt & 63;
            constexpr int NCH = BN * 17;                       // 16-byte chunks of a filter stage: 16 data + 1 pad per row
            constexpr int NPIECE = (NCH + 63) / 64;            // 1 KiB pieces (64 lanes x 16 bytes)
            constexpr int MAXP = (NPIECE + NDW - 1) / NDW;
            constexpr int NCHX = BM * 17;                      // the activation stage likewise (PF = 6)
            constexpr int NPX = XDMA ? (NCHX + 63) / 64 : 0;
            constexpr int MAXPX = XDMA ? (NPX + NDW - 1) / NDW : 1;
            const float *__restrict__ wg = p.w + (long)g * p.w_gs + (long)kt0 * BK;
            const float *src[MAXP];
            bool live[MAXP];
            int mine = 0;                                      // pieces this wave moves per k-tile (wave-uniform)
#pragma unroll
            for (int q = 0; q < MAXP; ++q) {
                const int piece = dwv + q * NDW, c = 64 * piece + lane;
                live[q] = piece < NPIECE && c < NCH;
                const int row = c / 17, ch = c - row * 17, co = bn0 + row;
                const bool ok = live[q] && ch < 16 && co < p.Cout;
                src[q] = wg + (ok ? (long)co * p.K + ch * 4 : 0L);      // pad chunks and rows past Cout fetch the bank's first bytes
                mine += piece < NPIECE ? 1 : 0;
            }
            // activation pieces: lane -> (pixel row of the tile, 16-byte chunk of its 64-channel block); the tap moves per k-tile
            const float *__restrict__ xg = p.x + (long)g * p.x_gs;
            int xih0[MAXPX], xiw0[MAXPX], xch[MAXPX];
            long xnb[MAXPX];
            bool xlive[MAXPX], xrow[MAXPX];
            if constexpr (XDMA) {
#pragma unroll
                for (int q = 0; q < MAXPX; ++q) {
                    const int piece = dwv + q * NDW, c = 64 * piece + lane;
                    xlive[q] = piece < NPX && c < NCHX;
                    const int row = c / 17, ch = c - row * 17, m = bm0 + row;
                    xrow[q] = xlive[q] && ch < 16 && m < p.M;          // a real pixel and a data chunk (else: zeros)
                    const int mm = xrow[q] ? m : 0;
                    const int n = mm / p.P, pix = mm - n * p.P;
                    const int oh = pix / p.OW, ow = pix - oh * p.OW;
                    xih0[q] = oh * p.stride - p.pad_h;
                    xiw0[q] = ow * p.stride - p.pad_w;
                    xnb[q] = (long)n * p.H * p.W * p.Cin;
                    xch[q] = ch * 4;
                    mine += piece < NPX ? 1 : 0;
                }
            }
            const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(smem + WBASE);
            const uint32_t ldsx = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
            auto dma = [&](const float *sp, uint32_t dst) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(sp), "s"(dst) : "memory");
            };
            auto issue = [&](int tile) {                       // k-tile `tile` of this workgroup's range -> stage tile % NSW
                const uint32_t stg = lds0 + (uint32_t)((tile % NSW) * BN * LD * 4);
#pragma unroll
                for (int q = 0; q < MAXP; ++q) {
                    const int piece = dwv + q * NDW;
                    if (piece < NPIECE && live[q])
                        dma(src[q] + (long)tile * BK, __builtin_amdgcn_readfirstlane(stg + (uint32_t)(piece * 1024)));
                }
                if constexpr (XDMA) {
                    const int kt = kt0 + tile, tap = kt / cch, cc = kt - tap * cch;
                    const int kh = tap / p.KW, kw = tap - kh * p.KW;
                    const uint32_t stx = ldsx + (uint32_t)((tile % NSX) * BM * LD * 4);
#pragma unroll
                    for (int q = 0; q < MAXPX; ++q) {
                        const int piece = dwv + q * NDW;
                        if (piece < NPX && xlive[q]) {
                            const int ih = xih0[q] + kh * p.dil_h, iw = xiw0[q] + kw * p.dil_w;
                            const bool in = xrow[q] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                            const float *sp = in ? xg + xnb[q] + ((long)ih * p.W + iw) * p.Cin + cc * BK + xch[q] : p.zero;
                            dma(sp, __builtin_amdgcn_readfirstlane(stx + (uint32_t)(piece * 1024)));
                        }
                    }
                }
            };
            // at most the pieces of the LA - 2 newest k-tiles may stay in flight (drain = true: none - the tail, where fewer exist)
            auto land = [&](bool drain) {
                if (drain) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
                switch (mine) {
                case 1: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * (LA - 2)) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (LA - 2)) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (LA - 2)) : "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (LA - 2)) : "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(5 * (LA - 2)) : "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 * (LA - 2)) : "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(7 * (LA - 2)) : "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * (LA - 2)) : "memory"); break;
                case 9: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(9 * (LA - 2)) : "memory"); break;
                case 10: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(10 * (LA - 2)) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                }
            };
            for (int j = 0; j < LA && j < nt; ++j) issue(j);
            land(nt < LA);                                     // k-tiles 0 and 1 are in LDS before the first barrier
            asm volatile("s_barrier" ::: "memory");
#ifdef USOT_TRACE
            unsigned *trd = (XDMA && p.ksplit == 1 && p.ws && bid0 == 0 && dt == 0) ? (unsigned *)p.ws : nullptr;
#define USOT_DSTAMP(slot, t) if (trd && (t) < 64) trd[(t) * 8 + (slot)] = (unsigned)__builtin_readcyclecounter()
#else
#define USOT_DSTAMP(slot, t)
#endif
            for (int t = 0; t < nt; ++t) {
                const bool more = t + LA < nt;
                USOT_DSTAMP(4, t);
                if (more) issue(t + LA);
                USOT_DSTAMP(3, t);
                USOT_DSTAMP(5, t);
                land(!more);                                   // k-tile t + 2 has landed when the consumers pass this step's barrier
                USOT_DSTAMP(6, t);
                asm volatile("s_barrier" ::: "memory");
                USOT_DSTAMP(7, t);
            }
#undef USOT_DSTAMP
            return;
        }
    }

    if (producer) {
        const float *__restrict__ xg = p.x + (long)g * p.x_gs;
        const float *__restrict__ wg = p.w + (long)g * p.w_gs;
        const int lr = tid / CPR, kc = tid % CPR;
        int x_ih0[XI], x_iw0[XI];
        long x_nb[XI];
        bool x_ok[XI];
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int row = lr + RPP * i;
            const int m = bm0 + row;
            x_ok[i] = (row < BM) && (m < p.M);
            const int mm = x_ok[i] ? m : 0;
            const int n = mm / p.P, pix = mm - n * p.P;
            const int oh = pix / p.OW, ow = pix - oh * p.OW;
            x_ih0[i] = oh * p.stride - p.pad_h;
            x_iw0[i] = ow * p.stride - p.pad_w;
            x_nb[i] = (long)n * p.H * p.W * p.Cin + kc * 4;
        }
        const float *wp[WI];
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int row = lr + RPP * i;
            const int co = bn0 + row;
            wp[i] = wg + (long)((row < BN && co < p.Cout) ? co : 0) * p.K + kc * 4 + (long)kt0 * BK;
        }
        f32x4 xr[D][XI], wr[D][WI];
        bool xz[D][XI];
        const float *xp[XI];
        bool xin[XI];
        int cur_tap = kt0 / cch, cur_cc = kt0 - cur_tap * cch;
        auto set_tap = [&](int tap) {
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const int dh = kh * p.dil_h, dw = kw * p.dil_w;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int ih = x_ih0[i] + dh, iw = x_iw0[i] + dw;
                xin[i] = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                xp[i] = xg + x_nb[i] + ((long)ih * p.W + iw) * p.Cin;
            }
        };
        set_tap(cur_tap);
        auto load_tile = [&](auto dc, bool advance) {
            constexpr int d = decltype(dc)::value;
            const int c0 = cur_cc * BK;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                // unconditional load (padding taps read pixel 0 of the tensor and are zeroed at the
                // LDS store): a branch around the load makes the compiler's vmcnt bookkeeping
                // conservative and every buffer's last store then drains ALL loads in flight
#ifdef USOT_ABL_NOLOAD     // scripts/ablate_kstep.py: timing builds with parts of the kernel removed
                xr[d][i] = f32x4{1.f, 1.f, 1.f, 1.f};
#else
                xr[d][i] = *(const f32x4 *)(xin[i] ? xp[i] + c0 : xg + kc * 4);
#endif
                xz[d][i] = xin[i];
            }
            if constexpr (!WDMA) {
#pragma unroll
            for (int i = 0; i < WI; ++i) {
#if defined(USOT_ABL_NOLOAD) || defined(USOT_ABL_NOW) || defined(USOT_ABL_NOWLOAD)   // NOW: the W operand never travels; NOWLOAD / NOWSTORE / NOWREAD: one leg of it removed
                wr[d][i] = f32x4{1.f, 1.f, 1.f, 1.f};
#else
                wr[d][i] = *(const f32x4 *)wp[i];
#endif
                wp[i] += advance ? BK : 0;
            }
            }
            if (advance && ++cur_cc == cch) {
                cur_cc = 0;
                set_tap(++cur_tap);
            }
        };
        auto store_tile = [&](auto dc, int st) {
            constexpr int d = decltype(dc)::value;
            float *sX = smem + st * XSTRIDE, *sW = smem + WBASE + st * WSTRIDE;
#ifdef USOT_ABL_NOSTORE
            return;
#endif
            // RPP is a multiple of 16 when BK = 64, so (row & 15) == (lr & 15) for every row of this thread
            const int kcs = (SWZ ? (kc ^ (lr & 15)) : kc) * 4;
#pragma unroll
            for (int i = 0; i < XI; ++i)
                if (BM % RPP == 0 || lr + RPP * i < BM) {
                    if constexpr (H16) {
                        // hi = fp16(8 x), lo = fp16(8 x - hi): halves kc*4 .. kc*4+3 of the row's hi plane and of its lo plane
                        const f32x4 v = (xz[d][i] ? xr[d][i] : f32x4{0.f, 0.f, 0.f, 0.f}) * 8.0f;
                        const uint32_t hi0 = usot_pack2_lp<true>(v[0], v[1]), hi1 = usot_pack2_lp<true>(v[2], v[3]);
                        const usot_f16x2 h0 = __builtin_bit_cast(usot_f16x2, hi0), h1 = __builtin_bit_cast(usot_f16x2, hi1);
                        const uint32_t lo0 = usot_pack2_lp<true>(v[0] - (float)h0[0], v[1] - (float)h0[1]);
                        const uint32_t lo1 = usot_pack2_lp<true>(v[2] - (float)h1[0], v[3] - (float)h1[1]);
                        const u32x2_t hi = {hi0, hi1}, lo = {lo0, lo1};
                        char *row = (char *)(sX + (lr + RPP * i) * LD);
                        *(u32x2_t *)(row + kc * 8) = hi;
                        *(u32x2_t *)(row + 128 + kc * 8) = lo;
                    } else {
                        *(f32x4 *)(sX + (lr + RPP * i) * LD + kcs) = xz[d][i] ? xr[d][i] : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#if !defined(USOT_ABL_NOW) && !defined(USOT_ABL_NOWSTORE)
            if constexpr (!WDMA) {
#pragma unroll
            for (int i = 0; i < WI; ++i)
                if (BN % RPP == 0 || lr + RPP * i < BN) *(f32x4 *)(sW + (lr + RPP * i) * LD + kcs) = wr[d][i];
            }
#elif defined(USOT_ABL_NOWSTORE)
#pragma unroll
            for (int i = 0; i < WI; ++i) {
                const f32x4 keep = wr[d][i];
                asm volatile("" :: "v"(keep));     // the loads stay, waited for here
            }
#endif
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, D >= 2 ? 1 : 0>;
        // Every load below is unconditional (past the last tile the pointers stop advancing and the
        // last tile is simply fetched again, never stored): the compiler can then count exactly
        // how many younger loads are in flight and waits with vmcnt(N > 0) instead of draining.
        // tiles 0 and 1 go to stages 0 and 1 before the first barrier; tiles 2 .. D+1 stay in
        // flight in register buffers 0 .. D-1 (tile t+2 lives in buffer t % D).  D up to 6: at batch 1 both
        // operands of a layer are cold (filters from the Infinity Cache / HBM once per frame, the activation
        // map from the other XCDs' write-backs), a round trip is 2-3 k-steps long.
        int lt = 0;                                   // next tile to fetch
        auto load_next = [&](auto dc) { load_tile(dc, lt + 1 < nt); ++lt; };
        if constexpr (D == 1) {
            load_next(I0{}); store_tile(I0{}, 0);
            load_next(I0{}); if (nt > 1) store_tile(I0{}, 1);
        } else {
            load_next(I0{});
            load_next(I1{});
            store_tile(I0{}, 0);
            if (nt > 1) store_tile(I1{}, 1);
        }
        auto each_buffer = [&](auto f) {              // f(buffer index constant) for buffers 0 .. D-1, in order
            [&]<int... Is>(std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
            (std::make_integer_sequence<int, D>{});
        };
        each_buffer([&](auto dc) { load_next(dc); });
        __syncthreads();
        int st2 = 2;                                  // stage that receives tile t+2
#ifdef USOT_TRACE   // scripts/trace_kstep.py: s_memtime stamps of one producer wave into the (unused) split-K workspace
        unsigned *trc = (p.ksplit == 1 && p.ws && bid0 == 0 && tid == 0) ? (unsigned *)p.ws : nullptr;
#define USOT_STAMP(slot, t) if (trc && (t) < 64) trc[(t) * 8 + (slot)] = (unsigned)__builtin_readcyclecounter()
#else
#define USOT_STAMP(slot, t)
#endif
#ifdef USOT_V3_STAGGER
        // the producer waves of a workgroup leave the k-step's barrier together and would all store their part of the tile at
        // once: 3 x NPW ds_write_b128 (13 LDS cycles each) in one burst in front of the consumers' fragment reads.  They have
        // ~300 cycles of slack per k-step (trace: barrier wait): wave w starts its step 64 x w x USOT_V3_STAGGER / 4 cycles later
        const int pw = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
        auto step = [&](auto dc, int t) {
            USOT_STAMP(4, t);
#ifdef USOT_V3_STAGGER
            for (int q = 0; q < pw * USOT_V3_STAGGER / 4; ++q) __builtin_amdgcn_s_sleep(1);
#endif
#ifdef USOT_TRACE
            {   // the wait for the oldest register buffer's loads, apart from the conversion / store instructions behind it
                constexpr int d_ = decltype(dc)::value;
                asm volatile("" :: "v"(xr[d_][XI - 1]));
                USOT_STAMP(3, t);
            }
#endif
            if (t + 2 < nt) store_tile(dc, st2);
            USOT_STAMP(5, t);
            load_next(dc);
            USOT_STAMP(6, t);
            st2 = st2 == 2 ? 0 : st2 + 1;
#ifndef USOT_ABL_NOBARRIER      // timing only: producers and consumers free-running (results are garbage)
            __syncthreads();
#endif
            USOT_STAMP(7, t);
        };
        int t = 0;
        for (; t + D <= nt; t += D) each_buffer([&](auto dc) { step(dc, t + decltype(dc)::value); });
        each_buffer([&](auto dc) {                    // tail: t is a multiple of D here, buffers continue 0, 1, ...
            if (t < nt) { step(dc, t); ++t; }
        });
        return;
    }

    // ---------------- consumers
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int l15 = lane & 15, quad = lane >> 4;
    const int fx_off = (wm * TM * 16 + l15) * LD + (SWZ ? 0 : quad * 4);
    const int fw_off = BM * LD + (wn * TN * 16 + l15) * LD + (SWZ ? 0 : quad * 4);
    // swizzled rows: chunk (4 r + quad) of row (16 b + l15) lives at chunk (4 r) ^ (quad ^ l15)
    const int swz = SWZ ? (quad ^ l15) * 4 : 0;     // one v_xor per round instead of four live offsets (80-VGPR budget: 2 workgroups / CU)
    constexpr int NSLOT = PF >= 2 ? NR : 2;
    static_assert(PF == 1 || (PF >= 2 && NR == 4), "PF >= 2 needs BK = 64");
    // split-fp16: the three products of one 32-k step on the blocks of this wave (slot ks = hi halves of k-step ks, slot ks + 2 = lo);
    // smallest terms first; a wave with a single block keeps the two cross terms on a second accumulator (no dependent chain of three)
    auto mma_h = [&](f32x4 (&a)[TN][TM], f32x4 &a2, const f32x4 (&whi)[TN], const f32x4 (&wlo)[TN], const f32x4 (&xhi)[TM], const f32x4 (&xlo)[TM], bool first) {
#ifdef USOT_ABL_NOMMA
        return;
#endif
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        auto h = [](const f32x4 &v) { return __builtin_bit_cast(f16x8_t, v); };
        if constexpr (TN * TM == 1) {
            a2      = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(wlo[0]), h(xhi[0]), first ? zero : a2, 0, 0, 0);
            a[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(whi[0]), h(xhi[0]), first ? zero : a[0][0], 0, 0, 0);
            a2      = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(whi[0]), h(xlo[0]), a2, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) a[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(wlo[i]), h(xhi[j]), first ? zero : a[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) a[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(whi[i]), h(xlo[j]), a[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) a[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(whi[i]), h(xhi[j]), a[i][j], 0, 0, 0);
        }
    };
    f32x4 fw[NSLOT][TN], fx[NSLOT][TM];
    auto read_frags = [&](int st, int r, int slot, int stw = 0) {       // stw: the filter stage (filter-DMA tiles only)
#ifdef USOT_ABL_NOREAD
        return;
#endif
        const float *base = smem + st * XSTRIDE + ((r * 16) ^ swz);
        const float *wbase = WDMA ? smem + WBASE - BM * LD + stw * WSTRIDE + r * 16 : base;     // (fw_off carries BM * LD)
#if defined(USOT_ABL_NOW) || defined(USOT_ABL_NOWREAD)
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[slot][i] = f32x4{1.f + st, 2.f, 3.f + r, 4.f};
#else
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[slot][i] = *(const f32x4 *)(wbase + fw_off + i * 16 * LD);
#endif
#pragma unroll
        for (int j = 0; j < TM; ++j) fx[slot][j] = *(const f32x4 *)(base + fx_off + j * 16 * LD);
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A wave with a single 16x16 block would chain every MFMA on one accumulator (40-cycle
    // dependent latency vs 32-cycle issue: an 80 % cap); it alternates two accumulators instead
    // (even / odd k-slots) and adds them at the end.
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
    // Blocked accumulation (see blocked_mma above): `acc` holds ONE k-tile's partial sums (its first MFMA takes C = 0),
    // `tot` the running total the finished k-tile is added to right after the barrier, when its MFMAs have long retired.
    BlockTotal tot[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) tot[i][j].clear();
    auto flush = [&]() {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) tot[i][j].add(acc[i][j]);
        if constexpr (TM * TN == 1) tot[0][0].add(acc2);
    };
    auto mma = [&](int slot, bool first) {
#ifdef USOT_ABL_NOMMA
        return;
#endif
        blocked_mma<TN, TM>(acc, acc2, fw[slot], fx[slot], first);
    };
    // bias and residual of this lane's outputs are fetched NOW, while the producers bring the first k-tiles:
    // loaded in the epilogue they add a dependent L2/HBM round trip to the tail of every layer, when no
    // other work is left to hide it.  (vectorised NHWC stores without split-K only.)
    const bool pre = p.vec_store && p.ksplit == 1;
    f32x4 pb[TN], pr[TN][TM];
    f32x4 psc[H16 ? TN : 1];                      // split-fp16: this lane's four output-channel scales per block, fetched up front too
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int co = bn0 + (wn * TN + i) * 16 + quad * 4;
        const bool cok = pre && co + 3 < p.Cout;
        pb[i] = (cok && p.bias) ? *(const f32x4 *)(p.bias + (long)g * p.b_gs + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (H16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) psc[i][e] = co + e < p.Cout ? p.wscale[(long)g * p.b_gs + co + e] : 0.f;      // (the bias' group stride: rows of the bank per group)
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = bm0 + (wm * TM + j) * 16 + l15;
            pr[i][j] = (cok && p.res && m < p.M)
                           ? *(const f32x4 *)(p.res + (long)g * p.r_gs + (long)m * p.res_cstride + p.res_coff + co)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
#ifdef USOT_TRACE
    unsigned *trc = (p.ksplit == 1 && p.ws && bid0 == 0 && tid == 0) ? (unsigned *)p.ws : nullptr;
#endif
    // PF = 3: the consumer's fragment traffic scheduled BY HAND.  hipcc's schedule of the plain loop (PF = 1) re-uses a fragment
    // register right behind the MFMA that read it and waits for that read at once (ds_read_b128 v[34:37] ... s_waitcnt lgkmcnt(1) ...
    // v_mfma ... v34), with s_nop 6-7 in front of reads that overwrite MFMA sources: the software pipeline of the source is gone
    // in the ISA.  Here the reads are inline asm (the compiler neither moves them behind their MFMAs nor waits for them), issued TWO
    // rounds ahead into one slot per round, and every round waits with a COUNTED s_waitcnt lgkmcnt(2 x reads per round) that carries
    // the round's fragment registers as operands, so no MFMA of the round can be scheduled above it.
    // (the registers travel as PARAMETERS: clang rejects asm operands that name captured arrays inside a generic lambda)
    auto aread_ = [](f32x4 (&w)[TN], f32x4 (&x)[TM], unsigned bw, unsigned bx, auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int LD = (BK == 64 ? BK + 4 : BK + 4);          // padded rows (the swizzled form is not built for PF = 3)
#pragma unroll
        for (int i = 0; i < TN; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[i]) : "v"(bw), "n"((i * 16 * LD + r * 16) * 4));
#pragma unroll
        for (int j = 0; j < TM; ++j)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[j]) : "v"(bx), "n"((j * 16 * LD + r * 16) * 4));
    };
    auto aread = [&](int st, auto rc, auto sc) {
        aread_(fw[decltype(sc)::value], fx[decltype(sc)::value], (unsigned)((st * STAGE + fw_off) * 4), (unsigned)((st * STAGE + fx_off) * 4), rc);
    };
    auto await_ = [](f32x4 (&w)[TN], f32x4 (&x)[TM], auto nc) {   // wait until at most nc reads are outstanding; ties the slot's registers to the wait
        static_assert(TN <= 2 && TM <= 2, "operand list below");
        if constexpr (TN == 2 && TM == 1) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(w[0]), "+v"(w[1]), "+v"(x[0]) : "n"(decltype(nc)::value));
        else if constexpr (TN == 1 && TM == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w[0]), "+v"(x[0]) : "n"(decltype(nc)::value));
        else if constexpr (TN == 1 && TM == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(w[0]), "+v"(x[0]), "+v"(x[1]) : "n"(decltype(nc)::value));
        else asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(x[0]), "+v"(x[1]) : "n"(decltype(nc)::value));
    };
    auto await_slot = [&](auto sc, auto nc) { await_(fw[decltype(sc)::value], fx[decltype(sc)::value], nc); };
    using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>; using R3 = std::integral_constant<int, 3>;
    int stw = 0;                                  // filter stage of k-tile t (filter-DMA tiles: t % NSW)
    if constexpr (H16) {
        if (nt > 0) { read_frags(0, 0, 0); read_frags(0, 2, 2); read_frags(0, 1, 1); read_frags(0, 3, 3); }
    } else if constexpr (PF == 3) {
        if (nt > 0) { aread(0, R0{}, R0{}); aread(0, R1{}, R1{}); }
    } else {
        if (nt > 0) read_frags(0, 0, 0);
        if (PF == 2 && nt > 0) read_frags(0, 1, 1);
    }
    int st = 0;
    for (int t = 0; t < nt; ++t) {
        const int st1 = st == 2 ? 0 : st + 1;
        USOT_STAMP(0, t);
        flush();                                      // the previous k-tile's block (zeros at t = 0)
        if constexpr (H16) {
            // tile t + 1 is complete in its stage since the barrier that ended step t - 1 (the producers run two tiles ahead): its
            // fragments are fetched as soon as a slot pair is free
            const bool more = t + 1 < nt;
            const int stw1 = stw == NSW - 1 ? 0 : stw + 1;
            const int sx1 = XDMA ? stw1 : st1;            // (PF = 6: the activation ring is as deep as the filter ring)
            mma_h(acc, acc2, fw[0], fw[2], fx[0], fx[2], true);
            if (more) { read_frags(sx1, 0, 0, stw1); read_frags(sx1, 2, 2, stw1); }
            mma_h(acc, acc2, fw[1], fw[3], fx[1], fx[3], false);
            if (more) { read_frags(sx1, 1, 1, stw1); read_frags(sx1, 3, 3, stw1); }
            st = st1;
            stw = stw1;
            USOT_STAMP(1, t);
            __syncthreads();
            USOT_STAMP(2, t);
            continue;
        }
        if constexpr (PF == 3) {
            constexpr int RPR = TN + TM;              // reads per round
            using W2 = std::integral_constant<int, 2 * RPR>;
            const bool more = t + 1 < nt;
            aread(st, R2{}, R2{});  await_slot(R0{}, W2{}); mma(0, true);
            aread(st, R3{}, R3{});  await_slot(R1{}, W2{}); mma(1, false);
            if (more) {
                aread(st1, R0{}, R0{}); await_slot(R2{}, W2{}); mma(2, false);
                aread(st1, R1{}, R1{}); await_slot(R3{}, W2{}); mma(3, false);
            } else {                                   // last k-tile: nothing left to fetch, the two newest rounds drain
                await_slot(R2{}, std::integral_constant<int, RPR>{}); mma(2, false);
                await_slot(R3{}, R0{}); mma(3, false);
            }
            st = st1;
            USOT_STAMP(1, t);
            asm volatile("s_barrier" ::: "memory");   // not __syncthreads(): its lgkmcnt(0) would drain the reads in flight for tile t + 1
            USOT_STAMP(2, t);
            continue;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if constexpr (PF == 2) {
                // tile t + 1 is complete in its stage since the barrier that ended step t - 1 (the producers run two tiles ahead)
                if (r + 2 < NR) read_frags(st, r + 2, r + 2);
                else if (t + 1 < nt) read_frags(st1, r + 2 - NR, r + 2 - NR);
                __builtin_amdgcn_sched_barrier(0);
                mma(r, r == 0);
            } else {
                if (r + 1 < NR) read_frags(st, r + 1, (r + 1) & 1);
                else if (t + 1 < nt) read_frags(st1, 0, 0);
                mma(r & 1, r == 0);
#ifdef USOT_V3_ILV
                // One fragment read BETWEEN two MFMAs instead of TN + TM reads in a burst in front of the round's MFMAs: a wave issues
                // in order and a ds_read_b128 holds its issue slot for ~30 cycles (trace: MFMA phase 1 396 cycles for 1 024 cycles of
                // MFMAs on the 32 x 64 tile, +90 per round whatever the prefetch distance) - under a running MFMA (32 cycles) that is free
#pragma unroll
                for (int q = 0; q < TN + TM; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * TN * TM - (TN + TM), 0);
#endif
            }
        }
        st = st1;
        USOT_STAMP(1, t);
#ifndef USOT_ABL_NOBARRIER
        __syncthreads();
#endif
        USOT_STAMP(2, t);
    }
#undef USOT_STAMP
    flush();
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = tot[i][j].get();
    if constexpr (H16) {
        // back to the unscaled sums (powers of two: exact) before the split-K slabs, the deferred reduction or the epilogue see them
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] *= psc[i];
        // Range contract, enforced: an activation beyond the fp16 window was staged as hi = inf, lo = -inf and every sum it entered is
        // NaN (inf - inf) by now; the epilogues below would turn that into a FINITE number (fmaxf(NaN, 0) = 0, exp(0) = 1).  Say so in
        // the caller's sticky word before bias / activation / the split-K slabs see the sums: the engine re-runs the frame on the
        // exact-fp32 tiles.  One class test per accumulator register, once per launch - nothing in the k-loop.
        if (p.ovf) {
            bool bad = false;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) bad |= !(__builtin_fabsf(acc[i][j][e]) <= 3.4028234664e38f);
            if (bad) __hip_atomic_store(p.ovf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    if (p.ksplit > 1) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = bm0 + (wm * TM + j) * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int co = bn0 + (wn * TN + i) * 16 + quad * 4;
                const long el = ((long)(ks * p.groups + g) * p.M + m) * p.Cout + co;
                if (co + 3 < p.Cout && (p.Cout & 3) == 0) {
                    ws_store4(p, el, acc[i][j]);
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Cout) ws_store1(p, el + e, acc[i][j][e]);
                }
            }
        }
        if (p.combine) splitk_combine<BM, BN>(p, g, t0, tiles, bm0, bn0, tid, (int *)smem);
        return;
    }
    const float *__restrict__ bg = p.bias ? p.bias + (long)g * p.b_gs : nullptr;
    const float *__restrict__ rg = p.res ? p.res + (long)g * p.r_gs : nullptr;
    float *__restrict__ yg = p.y + (long)g * p.y_gs;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = bm0 + (wm * TM + j) * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int co = bn0 + (wn * TN + i) * 16 + quad * 4;
            if (co >= p.Cout) continue;
            f32x4 v = acc[i][j];
            if (p.vec_store && co + 3 < p.Cout) {
                v += pb[i] + pr[i][j];                      // prefetched above (zeros where absent)
                const int a = co < p.act_split ? p.act : p.act2;
                if (a != USOT_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a);
                }
                if constexpr (H16) {
                    if (p.y_split) {
                        // the result leaves split (usot_conv_desc.y_split): per pixel and 64-channel block 64 hi halves then 64 lo halves of
                        // 8 x value - what a PF = 6 consumer's DMA copies into its LDS rows unchanged
                        const f32x4 w8 = v * 8.0f;
                        const uint32_t hi0 = usot_pack2_lp<true>(w8[0], w8[1]), hi1 = usot_pack2_lp<true>(w8[2], w8[3]);
                        const usot_f16x2 h0 = __builtin_bit_cast(usot_f16x2, hi0), h1 = __builtin_bit_cast(usot_f16x2, hi1);
                        const uint32_t lo0 = usot_pack2_lp<true>(w8[0] - (float)h0[0], w8[1] - (float)h0[1]);
                        const uint32_t lo1 = usot_pack2_lp<true>(w8[2] - (float)h1[0], w8[3] - (float)h1[1]);
                        char *blk = (char *)(yg + (long)m * p.y_cstride + p.y_coff + (co & ~63));
                        *(u32x2_t *)(blk + (co & 63) * 2) = u32x2_t{hi0, hi1};
                        *(u32x2_t *)(blk + 128 + (co & 63) * 2) = u32x2_t{lo0, lo1};
                        continue;
                    }
                }
                *(f32x4 *)(yg + (long)m * p.y_cstride + p.y_coff + co) = v;
            } else {
                const int n = m / p.P, pix = m - n * p.P;
                for (int e = 0; e < 4; ++e) {
                    const int c = co + e;
                    if (c >= p.Cout) break;
                    float s = v[e];
                    if (bg) s += bg[c];
                    if (rg) s += rg[(long)m * p.res_cstride + p.res_coff + c];
                    s = apply_act(s, c < p.act_split ? p.act : p.act2);
                    if (p.y_nchw) yg[((long)n * p.Cout + c) * p.P + pix] = s;
                    else          yg[(long)m * p.y_cstride + p.y_coff + c] = s;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// v3p: v3 as a PERSISTENT stream-K launch (whole-chip rounds).
//
// The launch's work is the list of (tile, k-tile) UNITS of all its problems - N-strip-major, so that a contiguous range of units
// is one filter strip x consecutive pixel tiles.  G resident workgroups (one or two per CU, never more than fit) each take
// a contiguous share of U / G units (XCD x takes the x-th eighth of the list: one filter strip per L2) and walk it SEGMENT by
// segment (a segment = the part of one tile inside the share).  A segment that covers its tile's whole k range ends in v3's
// epilogue.  A partial segment stores the wave's accumulators - in register order, 1 KiB per store, write-through (sc1) - to the
// tile's slab [part][wave][block][lane], drains, and takes a ticket PER CONSUMER WAVE; the wave that finds the other parts
// already there (the last to arrive) reads them back, adds them IN PART ORDER (= k order: the same bits whoever is last) and
// runs the epilogue.  No workgroup barrier, no spin, nothing that depends on placement or dispatch order (Guideline 16, recipe
// R1); tickets return to zero.  Producer waves never see any of this: they run ahead into the next segment's first k-tiles
// while the consumers finish the previous one - the launch pays ONE start-up and one drain instead of one per tile, and ends
// within one k-tile of U / G for every workgroup, which is what lets the 64 x 64 tile (2/64 of M N K instead of 3/64 or 4/64
// L2 bytes) be used where 552 tiles over 256 slots would otherwise quantise to three rounds.
struct SkInfo {                   // per problem: where its units start and how its tiles are numbered
    int ubase[5];                 // first unit of problem i (ubase[n] = U)
    int tbase[5];                 // first slab / ticket index of problem i
    int G;                        // workgroups of the launch
    int pmax;                     // parts a slab has room for
};

template <int BM, int BN, int WM, int WN, int D, int NPW, int BK = 64>
__global__ __launch_bounds__(256 + 64 * NPW, (BM * BN <= 32 * 64) ? (4 + NPW) / 2 : 1) void conv_igemm_f32_v3p(const ConvBatch bt, const SkInfo sk)
{
    static_assert(WM * WN == 4, "4 consumer wavefronts");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int LD = BK + 4;
    constexpr int CPR = BK / 4;
    constexpr int NPT = 64 * NPW;
    constexpr int RPP = NPT / CPR;
    constexpr int XI = (BM + RPP - 1) / RPP, WI = (BN + RPP - 1) / RPP;
    constexpr int NR = BK / 16;
    constexpr int STAGE = (BM + BN) * LD;
    constexpr int NBLK = TM * TN;                 // 16 x 16 blocks of a consumer wave

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool producer = threadIdx.x >= 256;
    const int tid = producer ? (int)threadIdx.x - 256 : (int)threadIdx.x;
    const int U = sk.ubase[bt.n];
    // XCD x (blocks b with b % 8 == x) walks the x-th contiguous eighth of the unit list
    const int wi = xcd_remap((int)blockIdx.x, sk.G);
    const int u_begin = (int)((long)U * wi / sk.G), u_end = (int)((long)U * (wi + 1) / sk.G);
    auto owner = [&](long u) { return (int)(((u + 1) * sk.G - 1) / U); };      // index (wi) of the workgroup whose share holds unit u

    auto each_buffer = [&](auto f) {
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }(std::make_integer_sequence<int, D>{});
    };

    int u = u_begin;
    while (u < u_end) {
        // ---- decode the segment [u, u1): problem, tile, k range (wave-uniform; both wave kinds compute the same)
        int pi = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (q < bt.n && u >= sk.ubase[q]) pi = q;
        const ConvK &p = bt.p[pi];
        const int KT = p.K / BK;
        const int cch = p.Cin / BK;
        const int local = u - sk.ubase[pi];
        const int tl = local / KT, kt0 = local - tl * KT;            // tile of the problem (groups x N-strips x M-tiles), first k-tile
        const int ut0 = sk.ubase[pi] + tl * KT;                       // the tile's first unit
        const int u1 = min(u_end, ut0 + KT);
        const int nt = u1 - u;
        const int tiles = p.MT * p.NT;
        const int g = tl / tiles, t0 = tl - g * tiles;
        const int bn0 = (t0 / p.MT) * BN, bm0 = (t0 % p.MT) * BM;
        u = u1;

        if (producer) {
            const float *__restrict__ xg = p.x + (long)g * p.x_gs;
            const float *__restrict__ wg = p.w + (long)g * p.w_gs;
            const int lr = tid / CPR, kc = tid % CPR;
            int x_ih0[XI], x_iw0[XI];
            long x_nb[XI];
            bool x_ok[XI];
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int row = lr + RPP * i;
                const int m = bm0 + row;
                x_ok[i] = (row < BM) && (m < p.M);
                const int mm = x_ok[i] ? m : 0;
                const int n = mm / p.P, pix = mm - n * p.P;
                const int oh = pix / p.OW, ow = pix - oh * p.OW;
                x_ih0[i] = oh * p.stride - p.pad_h;
                x_iw0[i] = ow * p.stride - p.pad_w;
                x_nb[i] = (long)n * p.H * p.W * p.Cin + kc * 4;
            }
            const float *wp[WI];
#pragma unroll
            for (int i = 0; i < WI; ++i) {
                const int row = lr + RPP * i;
                const int co = bn0 + row;
                wp[i] = wg + (long)((row < BN && co < p.Cout) ? co : 0) * p.K + kc * 4 + (long)kt0 * BK;
            }
            f32x4 xr[D][XI], wr[D][WI];
            bool xz[D][XI];
            const float *xp[XI];
            bool xin[XI];
            int cur_tap = kt0 / cch, cur_cc = kt0 - cur_tap * cch;
            auto set_tap = [&](int tap) {
                const int kh = tap / p.KW, kw = tap - kh * p.KW;
                const int dh = kh * p.dil_h, dw = kw * p.dil_w;
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    const int ih = x_ih0[i] + dh, iw = x_iw0[i] + dw;
                    xin[i] = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    xp[i] = xg + x_nb[i] + ((long)ih * p.W + iw) * p.Cin;
                }
            };
            set_tap(cur_tap);
            auto load_tile = [&](auto dc, bool advance) {
                constexpr int d = decltype(dc)::value;
                const int c0 = cur_cc * BK;
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    xr[d][i] = *(const f32x4 *)(xin[i] ? xp[i] + c0 : xg + kc * 4);
                    xz[d][i] = xin[i];
                }
#pragma unroll
                for (int i = 0; i < WI; ++i) {
                    wr[d][i] = *(const f32x4 *)wp[i];
                    wp[i] += advance ? BK : 0;
                }
                if (advance && ++cur_cc == cch) {
                    cur_cc = 0;
                    set_tap(++cur_tap);
                }
            };
            auto store_tile = [&](auto dc, int st) {
                constexpr int d = decltype(dc)::value;
                float *sX = smem + st * STAGE, *sW = sX + BM * LD;
#pragma unroll
                for (int i = 0; i < XI; ++i)
                    if (BM % RPP == 0 || lr + RPP * i < BM)
                        *(f32x4 *)(sX + (lr + RPP * i) * LD + kc * 4) = xz[d][i] ? xr[d][i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < WI; ++i)
                    if (BN % RPP == 0 || lr + RPP * i < BN) *(f32x4 *)(sW + (lr + RPP * i) * LD + kc * 4) = wr[d][i];
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, D >= 2 ? 1 : 0>;
            int lt = 0;
            auto load_next = [&](auto dc) { load_tile(dc, lt + 1 < nt); ++lt; };
            // the previous segment's last barrier has passed: every consumer has finished reading every stage
            load_next(I0{});
            load_next(I1{});
            store_tile(I0{}, 0);
            if (nt > 1) store_tile(I1{}, 1);
            each_buffer([&](auto dc) { load_next(dc); });
            __syncthreads();
            int st2 = 2;
            auto step = [&](auto dc, int t) {
                if (t + 2 < nt) store_tile(dc, st2);
                load_next(dc);
                st2 = st2 == 2 ? 0 : st2 + 1;
                __syncthreads();
            };
            int t = 0;
            for (; t + D <= nt; t += D) each_buffer([&](auto dc) { step(dc, t + decltype(dc)::value); });
            each_buffer([&](auto dc) {
                if (t < nt) { step(dc, t); ++t; }
            });
            continue;                                   // next segment: its loads are issued while the consumers finish this one
        }

        // ---------------- consumers
        const int lane = tid & 63, wave = tid >> 6;
        const int wm = wave % WM, wn = wave / WM;
        const int l15 = lane & 15, quad = lane >> 4;
        const int fx_off = (wm * TM * 16 + l15) * LD + quad * 4;
        const int fw_off = BM * LD + (wn * TN * 16 + l15) * LD + quad * 4;
        f32x4 fw[2][TN], fx[2][TM];
        auto read_frags = [&](int st, int r, int slot) {
            const float *base = smem + st * STAGE + r * 16;
#pragma unroll
            for (int i = 0; i < TN; ++i) fw[slot][i] = *(const f32x4 *)(base + fw_off + i * 16 * LD);
#pragma unroll
            for (int j = 0; j < TM; ++j) fx[slot][j] = *(const f32x4 *)(base + fx_off + j * 16 * LD);
        };
        f32x4 acc[TN][TM];
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
        BlockTotal tot[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; tot[i][j].clear(); }
        auto flush = [&]() {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) tot[i][j].add(acc[i][j]);
            if constexpr (TM * TN == 1) tot[0][0].add(acc2);
        };
        const bool whole = nt == KT;                     // the segment IS the tile: v3's epilogue
        // bias and residual travel while the producers bring the first k-tiles (whole tiles, vectorised stores)
        const bool pre = p.vec_store && whole;
        f32x4 pb[TN], pr[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int co = bn0 + (wn * TN + i) * 16 + quad * 4;
            const bool cok = pre && co + 3 < p.Cout;
            pb[i] = (cok && p.bias) ? *(const f32x4 *)(p.bias + (long)g * p.b_gs + co) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = bm0 + (wm * TM + j) * 16 + l15;
                pr[i][j] = (cok && p.res && m < p.M)
                               ? *(const f32x4 *)(p.res + (long)g * p.r_gs + (long)m * p.res_cstride + p.res_coff + co)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
        read_frags(0, 0, 0);
        int st = 0;
        for (int t = 0; t < nt; ++t) {
            const int st1 = st == 2 ? 0 : st + 1;
            flush();
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (r + 1 < NR) read_frags(st, r + 1, (r + 1) & 1);
                else if (t + 1 < nt) read_frags(st1, 0, 0);
                blocked_mma<TN, TM>(acc, acc2, fw[r & 1], fx[r & 1], r == 0);
            }
            st = st1;
            __syncthreads();
        }
        flush();
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = tot[i][j].get();

        if (!whole) {
            // ---- partial tile: this wave's blocks -> slab [tile][part][wave][block][lane]; one ticket per (tile, wave)
            const int w_first = owner(ut0), w_last = owner((long)ut0 + KT - 1);
            const int nparts = w_last - w_first + 1, part = wi - w_first;
            const long tslot = sk.tbase[pi] + tl;
            float *slab = p.ws + (tslot * sk.pmax) * (long)(BM * BN);
            int *tick = (int *)(p.ws + (long)sk.tbase[bt.n] * sk.pmax * (BM * BN)) + tslot * 4 + wave;
            const long ntot = (long)sk.tbase[bt.n] * sk.pmax * (BM * BN) * 4;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.ws, 0, (int)(ntot > 0x7fffffffL ? 0x7fffffffL : ntot), 0x00020000);
            auto off = [&](int prt, int blk) { return (int)((((tslot * sk.pmax + prt) * 4 + wave) * NBLK + blk) * 64 + lane) * 16; };
            (void)slab;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rs, off(part, i * TM + j), 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int ticket = 0;
            if (lane == 0) ticket = __hip_atomic_fetch_add(tick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket != nparts - 1) continue;           // another part's wave finishes this wave-tile
            if (lane == 0) __hip_atomic_store(tick, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // sum the parts in order (this wave's own part from the slab too: the same bits whichever wave is last)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    for (int q = 0; q < nparts; ++q)
                        v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off(q, i * TM + j), 0, 16));
                    acc[i][j] = v;
                }
        }
        const float *__restrict__ bg = p.bias ? p.bias + (long)g * p.b_gs : nullptr;
        const float *__restrict__ rg = p.res ? p.res + (long)g * p.r_gs : nullptr;
        float *__restrict__ yg = p.y + (long)g * p.y_gs;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = bm0 + (wm * TM + j) * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int co = bn0 + (wn * TN + i) * 16 + quad * 4;
                if (co >= p.Cout) continue;
                f32x4 v = acc[i][j];
                if (p.vec_store && co + 3 < p.Cout) {
                    if (whole) {
                        v += pb[i] + pr[i][j];
                    } else {
                        if (bg) v += *(const f32x4 *)(bg + co);
                        if (rg) v += *(const f32x4 *)(rg + (long)m * p.res_cstride + p.res_coff + co);
                    }
                    const int a = co < p.act_split ? p.act : p.act2;
                    if (a != USOT_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a);
                    }
                    *(f32x4 *)(yg + (long)m * p.y_cstride + p.y_coff + co) = v;
                } else {
                    const int n = m / p.P, pix = m - n * p.P;
                    for (int e = 0; e < 4; ++e) {
                        const int c = co + e;
                        if (c >= p.Cout) break;
                        float s2 = v[e];
                        if (bg) s2 += bg[c];
                        if (rg) s2 += rg[(long)m * p.res_cstride + p.res_coff + c];
                        s2 = apply_act(s2, c < p.act_split ? p.act : p.act2);
                        if (p.y_nchw) yg[((long)n * p.Cout + c) * p.P + pix] = s2;
                        else          yg[(long)m * p.y_cstride + p.y_coff + c] = s2;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// ws ("weight-streaming"): v3's producer / consumer split with the FILTER operand taken out of LDS.
//
// Measured on v3 (scripts/ablate_kstep.py, -DUSOT_ABL_NOW: the W operand neither loaded, staged nor read; 32 x 64 tile):
// Conf_Fusion's conv 112.8 -> 66.8 us (= the MFMA time of its 10.3 GFLOP at 2.4 GHz), one tower level 28.4 -> 18.3,
// layer3's 3x3 27.7 -> 18.0 — two thirds of a k-step's LDS traffic (16 of 24 KB written, 32 of 48 KB read) and two thirds
// of the producers' instructions are the filter tile, and that traffic, not the matrix pipe, is what a CU runs out of
// (two workgroups per CU are only 1.12 x faster than one).  Filters are constants: the host stores them in MFMA FRAGMENT
// order (usot_conv_pack_wfrag: [channel block of 16][k-tile of 64][round of 16 k][lane][4 floats], a permutation
// inside every 16-row block, so row offsets that are multiples of 16 and group strides keep their values) and each
// consumer wave fetches the A fragments of ITS 16 channels straight from global memory into registers, DW k-tiles
// ahead: one fully coalesced 1-KiB global_load_dwordx4 per round, no LDS, no second reader — the four consumer waves
// split the CHANNELS of the tile (wave w: channels 16 w TN ... ), every wave multiplies all BM pixels.  The activation
// tile keeps v3's path (producer waves, global -> registers -> LDS, three stages, one barrier per k-tile); a consumer
// reads TM = BM / 16 B fragments per round for TM x TN MFMA blocks.  All W loads are unconditional and the k-loop is
// unrolled over the DW register buffers, so the compiler counts vmcnt exactly (nothing else of a consumer's loop is
// a vector-memory instruction).
template <int BM, int BN, int D, int NPW, int DW, int WM = 1>
__global__ __launch_bounds__(256 + 64 * NPW) void conv_igemm_f32_ws(const ConvBatch bt)
{
    int pi = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < bt.n && (int)blockIdx.x >= bt.start[q]) pi = q;
    const ConvK &p = bt.p[pi];
    const int bid0 = (int)blockIdx.x - bt.start[pi];
    constexpr int BK = 64;
    constexpr int WN = 4 / WM;                    // consumer waves: WM along the pixels x WN along the channels
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    static_assert(TM >= 1 && TN >= 1 && BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "four consumer waves split the tile");
    constexpr int LD = BK;                        // unpadded rows, chunks XOR-swizzled by the row (see v3)
    constexpr int CPR = BK / 4;
    constexpr int NPT = 64 * NPW;
    constexpr int RPP = NPT / CPR;
    constexpr int XI = (BM + RPP - 1) / RPP;
    constexpr int NR = BK / 16;
    constexpr int STAGE = BM * LD;
    static_assert(RPP % 16 == 0, "a producer thread's rows share row & 15");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool producer = threadIdx.x >= 256;
    const int tid = producer ? (int)threadIdx.x - 256 : (int)threadIdx.x;
    const int tiles = p.MT * p.NT;
    const int total = tiles * p.groups * p.ksplit;
    const int b = xcd_remap(bid0, total);
    const int z = b / tiles, t0 = b - z * tiles;
    const int g = z / p.ksplit, ks = z - g * p.ksplit;
    const int bn0 = (t0 / p.MT) * BN, bm0 = (t0 % p.MT) * BM;
    const int KT = p.K / BK;
    const int cch = p.Cin / BK;
    const int kt0 = (int)((long)KT * ks / p.ksplit);
    const int kt1 = (int)((long)KT * (ks + 1) / p.ksplit);
    const int nt = kt1 - kt0;

    auto each_of = [&]<int... Is>(std::integer_sequence<int, Is...>, auto f) { (f(std::integral_constant<int, Is>{}), ...); };

    if (producer) {
        const float *__restrict__ xg = p.x + (long)g * p.x_gs;
        const int lr = tid / CPR, kc = tid % CPR;
        int x_ih0[XI], x_iw0[XI];
        long x_nb[XI];
        bool x_ok[XI];
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int row = lr + RPP * i;
            const int m = bm0 + row;
            x_ok[i] = (row < BM) && (m < p.M);
            const int mm = x_ok[i] ? m : 0;
            const int n = mm / p.P, pix = mm - n * p.P;
            const int oh = pix / p.OW, ow = pix - oh * p.OW;
            x_ih0[i] = oh * p.stride - p.pad_h;
            x_iw0[i] = ow * p.stride - p.pad_w;
            x_nb[i] = (long)n * p.H * p.W * p.Cin + kc * 4;
        }
        f32x4 xr[D][XI];
        bool xz[D][XI];
        const float *xp[XI];
        bool xin[XI];
        int cur_tap = kt0 / cch, cur_cc = kt0 - cur_tap * cch;
        auto set_tap = [&](int tap) {
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const int dh = kh * p.dil_h, dw = kw * p.dil_w;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int ih = x_ih0[i] + dh, iw = x_iw0[i] + dw;
                xin[i] = x_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                xp[i] = xg + x_nb[i] + ((long)ih * p.W + iw) * p.Cin;
            }
        };
        set_tap(cur_tap);
        auto load_tile = [&](auto dc, bool advance) {
            constexpr int d = decltype(dc)::value;
            const int c0 = cur_cc * BK;
#pragma unroll
            for (int i = 0; i < XI; ++i) {      // unconditional (see v3): padding taps read pixel 0 and are zeroed at the LDS store
                xr[d][i] = *(const f32x4 *)(xin[i] ? xp[i] + c0 : xg + kc * 4);
                xz[d][i] = xin[i];
            }
            if (advance && ++cur_cc == cch) {
                cur_cc = 0;
                set_tap(++cur_tap);
            }
        };
        auto store_tile = [&](auto dc, int st) {
            constexpr int d = decltype(dc)::value;
            float *sX = smem + st * STAGE;
            const int kcs = (kc ^ (lr & 15)) * 4;
#pragma unroll
            for (int i = 0; i < XI; ++i)
                if (BM % RPP == 0 || lr + RPP * i < BM)
                    *(f32x4 *)(sX + (lr + RPP * i) * LD + kcs) = xz[d][i] ? xr[d][i] : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, D >= 2 ? 1 : 0>;
        int lt = 0;
        auto load_next = [&](auto dc) { load_tile(dc, lt + 1 < nt); ++lt; };
        if constexpr (D == 1) {
            load_next(I0{}); store_tile(I0{}, 0);
            load_next(I0{}); if (nt > 1) store_tile(I0{}, 1);
        } else {
            load_next(I0{});
            load_next(I1{});
            store_tile(I0{}, 0);
            if (nt > 1) store_tile(I1{}, 1);
        }
        each_of(std::make_integer_sequence<int, D>{}, [&](auto dc) { load_next(dc); });
        __syncthreads();
        int st2 = 2;
        auto step = [&](auto dc, int t) {
            if (t + 2 < nt) store_tile(dc, st2);
            load_next(dc);
            st2 = st2 == 2 ? 0 : st2 + 1;
            __syncthreads();
        };
        int t = 0;
        for (; t + D <= nt; t += D) each_of(std::make_integer_sequence<int, D>{}, [&](auto dc) { step(dc, t + decltype(dc)::value); });
        each_of(std::make_integer_sequence<int, D>{}, [&](auto dc) {
            if (t < nt) { step(dc, t); ++t; }
        });
        return;
    }

    // ---------------- consumers: wave w owns channels bn0 + 16 TN w ... of all BM pixels
    const int lane = tid & 63, wave = (tid >> 6) / WM, wm = (tid >> 6) % WM;
    const int l15 = lane & 15, quad = lane >> 4;
    const int fx_off = (wm * TM * 16 + l15) * LD;
    const int swz = (quad ^ l15) * 4;
    // fragment-order filter stream of this wave's channel blocks (blocks past the padded bank read block 0: never stored)
    const int ncb = (p.Cout + 15) >> 4;
    const float *wq[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int cb = (bn0 >> 4) + wave * TN + i;
        wq[i] = p.w + (long)g * p.w_gs + ((long)(cb < ncb ? cb : 0) * KT + kt0) * (BK * 16) + lane * 4;
    }
    f32x4 wb[DW][TN][NR];
    f32x4 fx[2][TM];
    auto read_x = [&](int st, int r, int slot) {
        const float *base = smem + st * STAGE + ((r * 16) ^ swz) + fx_off;
#pragma unroll
        for (int j = 0; j < TM; ++j) fx[slot][j] = *(const f32x4 *)(base + j * 16 * LD);
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
    BlockTotal tot[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) tot[i][j].clear();
    auto flush = [&]() {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) tot[i][j].add(acc[i][j]);
        if constexpr (TM * TN == 1) tot[0][0].add(acc2);
    };
    // bias and residual first (v3: fetched while the first k-tiles travel), THEN the filter stream: vector loads return in
    // order, so the wait for the first filter fragments covers them
    const bool pre = p.vec_store && p.ksplit == 1;
    f32x4 pb[TN], pr[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int co = bn0 + (wave * TN + i) * 16 + quad * 4;
        const bool cok = pre && co + 3 < p.Cout;
        pb[i] = (cok && p.bias) ? *(const f32x4 *)(p.bias + (long)g * p.b_gs + co) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = bm0 + (wm * TM + j) * 16 + l15;
            pr[i][j] = (cok && p.res && m < p.M)
                           ? *(const f32x4 *)(p.res + (long)g * p.r_gs + (long)m * p.res_cstride + p.res_coff + co)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#ifdef USOT_WS_NOWAIT
    f32x4 sink[DW][TN][NR];
#endif
    int lt = 0;
    auto wnext = [&](auto dc) {                      // fetch the next k-tile of the stream into buffer dc (past the end: the last tile again)
        constexpr int d = decltype(dc)::value;
        const bool advance = lt + 1 < nt;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
#if defined(USOT_WS_NOWLOAD)          // timing builds (scripts/ablate_kstep.py): the filter stream never issued ...
                wb[d][i][r] = f32x4{1.f + r, 2.f, 3.f, 4.f + d};
#elif defined(USOT_WS_NOWAIT)         // ... or issued but never waited for inside the loop (MFMAs on constants)
                sink[d][i][r] = *(const f32x4 *)(wq[i] + r * 256);
                wb[d][i][r] = f32x4{1.f + r, 2.f, 3.f, 4.f + d};
#else
                wb[d][i][r] = *(const f32x4 *)(wq[i] + r * 256);
#endif
            }
            wq[i] += advance ? BK * 16 : 0;
        }
        ++lt;
    };
    each_of(std::make_integer_sequence<int, DW>{}, [&](auto dc) { wnext(dc); });
    __syncthreads();
    if (nt > 0) read_x(0, 0, 0);
    int st = 0;
    auto step = [&](auto dc, int t) {
        constexpr int d = decltype(dc)::value;
        const int st1 = st == 2 ? 0 : st + 1;
        flush();
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r + 1 < NR) read_x(st, r + 1, (r + 1) & 1);
            else if (t + 1 < nt) read_x(st1, 0, 0);
#ifndef USOT_WS_NOPIN
            // the NEXT round's B fragments leave LDS while this round's MFMAs issue; unpinned, hipcc sinks the reads below the
            // MFMAs, right in front of their first use, and every round then waits a full LDS latency (k-step 1 620 vs 1 040 cycles)
            __builtin_amdgcn_sched_barrier(0);
#endif
            f32x4 wfr[TN];
#pragma unroll
            for (int i = 0; i < TN; ++i) wfr[i] = wb[d][i][r];
            blocked_mma<TN, TM>(acc, acc2, wfr, fx[r & 1], r == 0);
        }
        wnext(dc);
        st = st1;
        __syncthreads();
    };
    {
        int t = 0;
        for (; t + DW <= nt; t += DW) each_of(std::make_integer_sequence<int, DW>{}, [&](auto dc) { step(dc, t + decltype(dc)::value); });
        each_of(std::make_integer_sequence<int, DW>{}, [&](auto dc) {
            if (t < nt) { step(dc, t); ++t; }
        });
    }
    flush();
#ifdef USOT_WS_NOWAIT
#pragma unroll
    for (int d = 0; d < DW; ++d)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int r = 0; r < NR; ++r) asm volatile("" :: "v"(sink[d][i][r]));
#endif
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = tot[i][j].get();

    if (p.ksplit > 1) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = bm0 + (wm * TM + j) * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int co = bn0 + (wave * TN + i) * 16 + quad * 4;
                const long el = ((long)(ks * p.groups + g) * p.M + m) * p.Cout + co;
                if (co + 3 < p.Cout && (p.Cout & 3) == 0) {
                    ws_store4(p, el, acc[i][j]);
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Cout) ws_store1(p, el + e, acc[i][j][e]);
                }
            }
        }
        if (p.combine) splitk_combine<BM, BN>(p, g, t0, tiles, bm0, bn0, tid, (int *)smem);
        return;
    }
    const float *__restrict__ bg = p.bias ? p.bias + (long)g * p.b_gs : nullptr;
    const float *__restrict__ rg = p.res ? p.res + (long)g * p.r_gs : nullptr;
    float *__restrict__ yg = p.y + (long)g * p.y_gs;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = bm0 + (wm * TM + j) * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int co = bn0 + (wave * TN + i) * 16 + quad * 4;
            if (co >= p.Cout) continue;
            f32x4 v = acc[i][j];
            if (p.vec_store && co + 3 < p.Cout) {
                v += pb[i] + pr[i][j];
                const int a = co < p.act_split ? p.act : p.act2;
                if (a != USOT_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a);
                }
                *(f32x4 *)(yg + (long)m * p.y_cstride + p.y_coff + co) = v;
            } else {
                const int n = m / p.P, pix = m - n * p.P;
                for (int e = 0; e < 4; ++e) {
                    const int c = co + e;
                    if (c >= p.Cout) break;
                    float s2 = v[e];
                    if (bg) s2 += bg[c];
                    if (rg) s2 += rg[(long)m * p.res_cstride + p.res_coff + c];
                    s2 = apply_act(s2, c < p.act_split ? p.act : p.act2);
                    if (p.y_nchw) yg[((long)n * p.Cout + c) * p.P + pix] = s2;
                    else          yg[(long)m * p.y_cstride + p.y_coff + c] = s2;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// wstat ("weight-stationary"): the big-K convolutions of the batch-1 frame with the FILTERS IN REGISTERS.
//
// What a v3 k-step waits for (round 5; scripts/ablate_kstep.py on Conf_Fusion's conv, 32 x 64 tile, two workgroups per CU):
// whole kernel 115 us; without the filter tile's global loads 111, without its LDS stores 109, WITHOUT THE CONSUMERS' TWO
// FILTER-FRAGMENT ds_read_b128 PER ROUND 77, with no filter traffic at all 67 (= the MFMA time of 10.3 GFLOP at 2.4 GHz).
// Neither deeper fragment prefetch (PF = 2) nor conflict-free rows (USOT_V3_SWZ) nor filters fetched straight from global
// memory (ws tiles) recover it: a consumer wave runs at the matrix pipe's rate only with ONE operand fragment read per
// 8 MFMAs.  Filters are constants, so they can stay where the MFMA reads them:
//   * a workgroup = 8 wavefronts (two per SIMD, all of them loaders AND consumers, 512 threads, one workgroup per CU) owns
//     32 output channels for a RANGE of 32-pixel tiles; wave j keeps the A fragments of those 32 channels for the k-rounds
//     j, j + 8, j + 16, ... (a round = 16 k) - K / 128 rounds x 2 channel blocks x 4 registers = 144 VGPRs at K = 2304 -
//     fetched ONCE per workgroup from the fragment-order bank (usot_conv_pack_wfrag_f32), every wave-load 1 KiB contiguous;
//   * the activation operand streams through LDS in panels of 32 pixels x 128 k (one round per wave and panel; 16 KB, three
//     stages, global -> registers -> LDS by all 512 threads two panels ahead, one barrier per panel): per panel a wave reads
//     TWO B fragments and issues 16 MFMAs (2 pixel blocks x 2 channel blocks x 4) - one read per 8 MFMAs;
//   * k is split over the eight waves INSIDE the workgroup: at the end of a pixel tile the eight partial 32 x 32 tiles meet
//     in LDS (32 KB; written before the panel's barrier, summed in wave order 0..7 by waves 0-3 - one 16 x 16 block each -
//     which add bias / residual / activation and store 16 bytes per lane) while the panel stream of the next tile is
//     already in flight.  Accumulation: a wave's chain is K / 8 products (288 at K = 2304, flushed into a running total
//     every four panels = 64 products like every other kernel of this file), then 8 partials in fixed order.
// Per CU and panel: 16 KB from L2 for 2 x 1 024 MFMA cycles per SIMD (v3: 48 KB), filters never re-read.  Needs
// Cin % 128 == 0 (a panel lies inside one filter tap), K == NST * 128 * RPS, Cout % 32 == 0.
template <int NST, int RPS, bool BLOCKED = false>
__global__ __launch_bounds__(512) void conv_wstat_f32(const ConvBatch bt)
{
    int pi = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < bt.n && (int)blockIdx.x >= bt.start[q]) pi = q;
    const ConvK &p = bt.p[pi];
    const int bid0 = (int)blockIdx.x - bt.start[pi];
    constexpr int BM = 32;
    constexpr int PK = 128 * RPS;                 // k per panel
    constexpr int LDP = PK + 4;
    constexpr int CPR = PK / 4;                   // 16-byte chunks per panel row
    constexpr int RPP = 512 / CPR;                // rows per pass of the 512 loader threads
    constexpr int XI = BM / RPP;
    constexpr int STAGE = BM * LDP;
    constexpr int NSTG = 4;                       // LDS stages: panel q + 1 is complete while panel q is multiplied
    static_assert(BM % RPP == 0, "loader passes");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *red = smem + NSTG * STAGE;             // [8 waves][4 blocks][64 lanes] f32x4

    // the problem's fields in registers (p is a dynamically indexed reference into the kernel arguments: every use would be an s_load)
    const int pM = p.M, pP = p.P, pOW = p.OW, pH = p.H, pW = p.W, pCin = p.Cin, pKW = p.KW;
    const int pstride = p.stride, ppad_h = p.pad_h, ppad_w = p.pad_w, pdil_h = p.dil_h, pdil_w = p.dil_w;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, quad = lane >> 4;
    const int units = p.groups * p.NT * p.pr;
    const int u = xcd_remap(bid0, units);
    const int pr = u % p.pr, gc = u / p.pr;
    const int cg = gc % p.NT, g = gc / p.NT;
    const int tile0 = (int)((long)p.MT * pr / p.pr), tile1 = (int)((long)p.MT * (pr + 1) / p.pr);
    const int ntile = tile1 - tile0;
    if (ntile <= 0) return;
    const int npan = ntile * NST;
    const int KT = p.K / 64;

    // ---- loader state (runs three panels ahead of the MFMAs); 32-bit element offsets (the launcher checks the map's size)
    const float *__restrict__ xg = p.x + (long)g * p.x_gs;
    const int lr = tid / CPR, kc = tid % CPR;
    int x_ih0[XI], x_iw0[XI], x_nb[XI];
    unsigned xo[XI];                              // BYTE offset of the row's source at the current tap; 0xffffffff: padding / past M
    // Loads go through a buffer descriptor over this group's input map: an offset past its end returns ZEROS, so padding taps and
    // the pixels past M need no select on the loaded data and no second pointer; the channel offset of the panel rides in the
    // instruction's scalar offset - a panel's two loads per thread cost no vector ALU instruction at all
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)xg, 0, (int)((long)p.N * pH * pW * pCin * 4), 0x00020000);
    int l_tile = tile0, l_s = 0, l_tap = 0, l_cc = 0;
    auto set_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int m = tile * BM + lr + RPP * i;
            const bool ok = m < pM;
            const int mm = ok ? m : 0;
            const int n = mm / pP, pix = mm - n * pP;
            const int oh = pix / pOW, ow = pix - oh * pOW;
            x_ih0[i] = ok ? oh * pstride - ppad_h : -(1 << 20);      // far outside: every tap fails the bounds test
            x_iw0[i] = ow * pstride - ppad_w;
            x_nb[i] = n * pH * pW * pCin + kc * 4;
        }
    };
    auto set_tap = [&](int tap) {
        const int kh = tap / pKW, kw = tap - kh * pKW;
        const int dh = kh * pdil_h, dw = kw * pdil_w;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int ih = x_ih0[i] + dh, iw = x_iw0[i] + dw;
            const bool in = (unsigned)ih < (unsigned)pH && (unsigned)iw < (unsigned)pW;
            xo[i] = in ? (unsigned)(x_nb[i] + (ih * pW + iw) * pCin) * 4u : 0xffffffffu;
        }
    };
    set_tile(l_tile);
    set_tap(0);
    f32x4 xr[2][XI];
    int l_left = npan;                                       // panels not yet fetched
    auto load_issue = [&](auto dc) {
        constexpr int d = decltype(dc)::value;
#pragma unroll
        for (int i = 0; i < XI; ++i)
#if defined(USOT_WSTAT_LOADSAME)     // timing: every panel fetches the SAME lines (L1 hits: the issue side of the loads alone)
            xr[d][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(kc * 16 + i * 4096 + lr * 512), 0, 0));
#elif defined(USOT_WSTAT_LOADHALF)   // timing: half the bytes (both loads of a thread fetch its first row)
            xr[d][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)xo[0], l_cc * 4, 0));
#else
            xr[d][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)xo[i], l_cc * 4, 0));
#endif
    };
    auto load_advance = [&]() {
        // wave-uniform counters: kept in scalar registers (readfirstlane: hipcc otherwise carries them in VGPRs under exec masks)
        l_left = __builtin_amdgcn_readfirstlane(l_left - 1);
        if (l_left > 0) {                   // past the last panel the state stops advancing (the panel is fetched again, never stored)
            l_cc = __builtin_amdgcn_readfirstlane(l_cc + PK);
            l_s = __builtin_amdgcn_readfirstlane(l_s + 1);
            if (l_s == NST) {
                l_s = 0; l_cc = 0; l_tap = 0;
                l_tile = __builtin_amdgcn_readfirstlane(l_tile + 1);
                set_tile(l_tile);
                set_tap(0);
            } else if (l_cc == pCin) {
                l_cc = 0;
                l_tap = __builtin_amdgcn_readfirstlane(l_tap + 1);
                set_tap(l_tap);
            }
        }
    };
    auto load_next = [&](auto dc) { load_issue(dc); load_advance(); };
    auto store_panel = [&](auto dc, int st) {
        constexpr int d = decltype(dc)::value;
        float *sX = smem + st * STAGE;
#ifdef USOT_WSTAT_NOWAIT        // timing: the fetched panel is never waited for (the stores write other registers)
#pragma unroll
        for (int i = 0; i < XI; ++i) *(f32x4 *)(sX + (lr + RPP * i) * LDP + kc * 4) = f32x4{1.f * lr, 2.f, 3.f * d, 4.f};
#else
#pragma unroll
        for (int i = 0; i < XI; ++i) *(f32x4 *)(sX + (lr + RPP * i) * LDP + kc * 4) = xr[d][i];
#endif
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // panels 0 and 1 go to LDS before the first barrier, panel 2 waits in register buffer 0, panel 3 in buffer 1
    load_next(I0{});
    load_next(I1{});

    // ---- this wave's filter fragments: rounds (s * 8 + wave) * RPS + rr of the 32 channels [cg * 32, cg * 32 + 32)
    f32x4 wb[NST][RPS][2];
    {
        const float *wbase = p.w + (long)g * p.w_gs + lane * 4;
#pragma unroll
        for (int s = 0; s < NST; ++s)
#pragma unroll
            for (int rr = 0; rr < RPS; ++rr) {
                const int R = (s * 8 + wave) * RPS + rr;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    wb[s][rr][tn] = *(const f32x4 *)(wbase + (((long)(cg * 2 + tn) * KT + (R >> 2)) * 4 + (R & 3)) * 256);
            }
    }
    store_panel(I0{}, 0);
    load_next(I0{});
    if (npan > 1) store_panel(I1{}, 1);
    load_next(I1{});
    __syncthreads();

    f32x4 acc[2][2];
    BlockTotal tot[BLOCKED ? 2 : 1][BLOCKED ? 2 : 1];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            acc[tn][tm] = zero;
            if constexpr (BLOCKED) tot[tn][tm].clear();
        }
    constexpr int FLUSH = RPS == 1 ? 4 : 2;       // BLOCKED: panels per 64-product block of a wave's chain
    const int fx_off = l15 * LDP + wave * RPS * 16 + quad * 4;
    // gap (0..3 = after MFMA group 0..2, 3 = before group 0) in which this wave stores / fetches: the store of buffer d must
    // precede the fetch into buffer d within a panel, so st_gap comes first in the order 3, 0, 1, 2
    const int wsl = __builtin_amdgcn_readfirstlane(wave & 3);
    const int st_gap = (wsl == 0 || wsl == 3) ? 3 : wsl - 1;  // program order of the gaps: 3, 0, 1, 2
    const int ld_gap = wsl == 3 ? 2 : wsl;
    f32x4 fx[2][RPS][2];                          // [slot][round][pixel block]: slot q & 1 holds panel q's B fragments
    auto read_x = [&](auto sl, int st) {
        constexpr int slot = decltype(sl)::value;
        const float *sX = smem + st * STAGE + fx_off;
#pragma unroll
        for (int rr = 0; rr < RPS; ++rr)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) fx[slot][rr][tm] = *(const f32x4 *)(sX + tm * 16 * LDP + rr * 16);
    };
    read_x(I0{}, 0);

    int q = 0, st = 0;                            // panel index, its LDS stage
    auto panel = [&](auto sc, auto dc, int tile) {
        constexpr int s = decltype(sc)::value;
        constexpr int d = decltype(dc)::value;    // q & 1: fragment slot of panel q, register buffer of panel q + 2 (stored now) and q + 4 (fetched now)
        const int st1 = (st + 1) & 3, st2 = (st + 2) & 3;
        // panel q + 1 has been complete in its stage since the previous barrier: its fragments travel while this panel is multiplied
#ifndef USOT_WSTAT_NOREAD
        if (q + 1 < npan) read_x(std::integral_constant<int, 1 - d>{}, st1);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BLOCKED && s % FLUSH == 0 && s > 0) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) tot[tn][tm].add(acc[tn][tm]);
        }
        // A wave issues in order and an MFMA waits for the pipe (32 cycles each, 64 with the SIMD's other wave): everything else
        // of the panel - the LDS store of panel q + 2, the fetch of panel q + 4, the loader's index arithmetic - is placed BETWEEN
        // the four k-slot groups of the panel's MFMAs (pinned: hipcc otherwise issues the 16 MFMAs first and the rest behind them,
        // where both waves of a SIMD sit in the same phase and the matrix pipe idles: 1 920 vs 1 024 cycles per panel)
        auto mma4 = [&](auto cc, int rr) {
            constexpr int c = decltype(cc)::value;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[s][rr][tn][c], fx[d][rr][tm][c],
                                                                       (((BLOCKED && s % FLUSH == 0) || s == 0) && rr == 0 && c == 0) ? zero : acc[tn][tm], 0, 0, 0);
        };
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;
        // ... and STAGGERED over the waves: all eight run the same panel between the same two barriers, and eight fetches (or eight
        // LDS stores) issued at the same point queue up in one unit while every wave's MFMA stream waits behind its own
        // instruction; wave w fetches in gap (w & 3) and stores in gap ((w + 2) & 3) of the panel's four MFMA groups
        auto gap = [&](int g4) {
            __builtin_amdgcn_sched_barrier(0);
#ifndef USOT_WSTAT_NOSTORE
            // panel q + 2 (fetched two panels ago) -> its stage (held panel q - 2, whose last reader passed two barriers ago)
            if (g4 == st_gap && q + 2 < npan) store_panel(dc, st2);
#endif
#ifndef USOT_WSTAT_NOLOAD
            if (g4 == ld_gap) load_issue(dc);                 // panel q + 4 (after the store of the same buffer: gaps are cyclic)
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        static_assert(RPS == 1, "the gap schedule below is written for one round per panel");
        gap(3);                                               // cyclic order of a buffer's uses: store (gap st) ... fetch (gap ld) ...
        mma4(C0{}, 0);
        gap(0);
        mma4(C1{}, 0);
        gap(1);
        mma4(C2{}, 0);
        gap(2);
        load_advance();
        __builtin_amdgcn_sched_barrier(0);
        mma4(C3{}, 0);
        if constexpr (s == NST - 1) {             // the tile's partial sums meet in LDS (read after this panel's barrier)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    f32x4 v = acc[tn][tm];
                    if constexpr (BLOCKED) { tot[tn][tm].add(v); v = tot[tn][tm].get(); tot[tn][tm].clear(); }
                    *(f32x4 *)(red + ((wave * 4 + tn * 2 + tm) * 64 + lane) * 4) = v;
                }
        }
        st = st1;
        ++q;
#ifndef USOT_WSTAT_NOBAR       // timing builds (results are garbage): no per-panel barrier / no panel stores / no panel loads / no fragment reads
        __syncthreads();
#endif
        if constexpr (s == NST - 1) {
            if (wave < 4) {                       // wave w: block (tn, tm) = (w >> 1, w & 1), the eight partials in wave order
                const int tn = wave >> 1, tm = wave & 1;
                f32x4 v = zero;
#pragma unroll
                for (int j = 0; j < 8; ++j) v += *(const f32x4 *)(red + ((j * 4 + wave) * 64 + lane) * 4);
                const int m = tile * BM + tm * 16 + l15;
                const int co = cg * 32 + tn * 16 + quad * 4;
                if (m < p.M) {
                    const float *__restrict__ bg = p.bias ? p.bias + (long)g * p.b_gs : nullptr;
                    const float *__restrict__ rg = p.res ? p.res + (long)g * p.r_gs : nullptr;
                    float *__restrict__ yg = p.y + (long)g * p.y_gs;
                    if (p.vec_store) {
                        if (bg) v += *(const f32x4 *)(bg + co);
                        if (rg) v += *(const f32x4 *)(rg + (long)m * p.res_cstride + p.res_coff + co);
                        const int a = co < p.act_split ? p.act : p.act2;
                        if (a != USOT_ACT_NONE) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a);
                        }
                        *(f32x4 *)(yg + (long)m * p.y_cstride + p.y_coff + co) = v;
                    } else {
                        const int n = m / p.P, pix = m - n * p.P;
                        for (int e = 0; e < 4; ++e) {
                            const int c = co + e;
                            float s2 = v[e];
                            if (bg) s2 += bg[c];
                            if (rg) s2 += rg[(long)m * p.res_cstride + p.res_coff + c];
                            s2 = apply_act(s2, c < p.act_split ? p.act : p.act2);
                            if (p.y_nchw) yg[((long)n * p.Cout + c) * p.P + pix] = s2;
                            else          yg[(long)m * p.y_cstride + p.y_coff + c] = s2;
                        }
                    }
                }
            }
        }
    };
#ifdef USOT_WSTAT_NOWAIT
#define USOT_WSTAT_SINK() do { _Pragma("unroll") for (int i = 0; i < XI; ++i) { asm volatile("" :: "v"(xr[0][i])); asm volatile("" :: "v"(xr[1][i])); } } while (0)
#else
#define USOT_WSTAT_SINK()
#endif
    for (int tile = tile0; tile < tile1; ++tile) {
        const bool flip = (NST & 1) && ((tile - tile0) & 1);       // odd panel counts: the buffer parity flips from tile to tile
        if (!flip) {
            [&]<int... Is>(std::integer_sequence<int, Is...>) {
                (panel(std::integral_constant<int, Is>{}, std::integral_constant<int, Is & 1>{}, tile), ...);
            }(std::make_integer_sequence<int, NST>{});
        } else {
            [&]<int... Is>(std::integer_sequence<int, Is...>) {
                (panel(std::integral_constant<int, Is>{}, std::integral_constant<int, (Is + 1) & 1>{}, tile), ...);
            }(std::make_integer_sequence<int, NST>{});
        }
    }
    USOT_WSTAT_SINK();
}

struct TileCfg { int bm, bn, bk, stages, ksw; void (*fn)(const ConvBatch); int threads; int depth; int wfrag; int dw; int nst = 0; int rps = 0;
                 void (*skfn)(const ConvBatch, const SkInfo) = nullptr; };

#define TILE(bm, bn, wm, wn) { bm, bn, 32, 2, 1, conv_igemm_f32<bm, bn, wm, wn>, 256, 1, 0, 0 }
#define TILE2(bm, bn, wm, wn, bk) { bm, bn, bk, 3, 1, conv_igemm_f32_v2<bm, bn, wm, wn, bk>, 256, 1, 0, 0 }
#define TILE3(bm, bn, wm, wn, bk, ksw) { bm, bn, bk, 3, ksw, conv_igemm_f32_v2<bm, bn, wm, wn, bk, ksw>, 256 * ksw, 1, 0, 0 }
#define TILE4(bm, bn, wm, wn, bk) { bm, bn, bk, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, bk>, 512, 1, 0, 0 }
#define TILE10(bm, bn, wm, wn, bk, d, npw) { bm, bn, bk, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, bk, d, npw>, 256 + 64 * npw, d, 0, 0 }
#define TILEW(bm, bn, d, npw, dw) { bm, bn, 64, 3, 1, conv_igemm_f32_ws<bm, bn, d, npw, dw>, 256 + 64 * npw, d, 1, dw }
#define TILEW2(bm, bn, d, npw, dw) { bm, bn, 64, 3, 1, conv_igemm_f32_ws<bm, bn, d, npw, dw, 2>, 256 + 64 * npw, d, 1, dw }
#define TILE11(bm, bn, wm, wn, bk, d, npw) { bm, bn, bk, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, bk, d, npw, 2>, 256 + 64 * npw, d, 0, 2 }
#define TILESB(nst, rps) { 32, 32, 64, 3, 1, conv_wstat_f32<nst, rps, true>, 512, 2, 1, 0, nst, rps }
#define TILEP(bm, bn, wm, wn, d, npw) { bm, bn, 64, 3, 1, nullptr, 256 + 64 * npw, d, 0, 0, 0, 0, conv_igemm_f32_v3p<bm, bn, wm, wn, d, npw> }
#define TILEP32(bm, bn, wm, wn, d, npw) { bm, bn, 32, 3, 1, nullptr, 256 + 64 * npw, d, 0, 0, 0, 0, conv_igemm_f32_v3p<bm, bn, wm, wn, d, npw, 32> }
#define TILES(nst, rps) { 32, 32, 64, 3, 1, conv_wstat_f32<nst, rps>, 512, 2, 1, 0, nst, rps }
#define TILE12(bm, bn, wm, wn, bk, d, npw) { bm, bn, bk, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, bk, d, npw>, 64 * wm * wn + 64 * npw, d, 0, 8 }
#define TILE13(bm, bn, wm, wn, bk, d, npw) { bm, bn, bk, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, bk, d, npw, 3>, 64 * wm * wn + 64 * npw, d, 0, 3 }
#define TILEH(bm, bn, wm, wn, d, npw) { bm, bn, 64, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, 64, d, npw, 4>, 64 * wm * wn + 64 * npw, d, 2, 4 }
#define TILEHD(bm, bn, wm, wn, d, npw) { bm, bn, 64, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, 64, d, npw, 5>, 64 * wm * wn + 64 * npw + 256, d, 2, 5 }
#define TILEHX(bm, bn, wm, wn) { bm, bn, 64, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, 64, 2, 4, 6>, 64 * wm * wn + 768, 2, 2, 6 }
#define TILE5(bm, bn, wm, wn, bk, d) { bm, bn, bk, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, bk, d>, 512, d, 0, 0 }
// The DEFAULT build compiles the ROUTED tiles only: the ids the tuning tables (usot_amd/data/tuning_gfx950.json, tuning_split16_gfx950.json),
// engine.SPLIT16_TILES, the engine's deferred-launch options and pick_tile() below can select (tests/test_abi_and_build.py asserts the
// two sets are equal).  Every other id - the experiments the lab notebook records as "parity-green, measured slower, not routed" - keeps
// its NUMBER but is an empty slot unless the library is built with -DUSOT_EXPERIMENTS (usot_amd/build.py: USOT_EXPERIMENTS=1); the
// launcher answers USOT_ENOTBUILT for it and usot_conv_tile_built() says so up front.
#ifdef USOT_EXPERIMENTS
#define XT(...) __VA_ARGS__
#else
#define XT(...) TileCfg{0, 0, 0, 0, 0, nullptr, 0, 0, 0, 0}
#endif
const TileCfg kTiles[] = {
    TILE(128, 128, 2, 2),   // 1: batched backbone
    TILE(128, 64, 2, 2),    // 2
    XT(TILE(64, 128, 2, 2)),    // 3
    TILE(64, 64, 2, 2),     // 4
    TILE(32, 64, 2, 2),     // 5
    XT(TILE(64, 32, 2, 2)),     // 6
    TILE(32, 32, 2, 2),     // 7
    TILE(16, 64, 1, 4),     // 8: tiny M (template-side encoders)
    XT(TILE(16, 128, 1, 4)),    // 9
    XT(TILE(32, 128, 2, 2)),    // 10
    XT(TILE2(128, 128, 2, 2, 32)),  // 11: v2 (3-stage, mid-tile barrier)
    XT(TILE2(64, 64, 2, 2, 32)),    // 12
    XT(TILE2(64, 64, 2, 2, 64)),    // 13
    TILE2(32, 64, 2, 2, 32),    // 14
    TILE2(32, 64, 2, 2, 64),    // 15
    TILE2(32, 32, 2, 2, 64),    // 16
    XT(TILE2(64, 128, 2, 2, 32)),   // 17
    XT(TILE2(128, 64, 2, 2, 32)),   // 18
    XT(TILE2(32, 128, 2, 2, 32)),   // 19
    TILE2(16, 64, 1, 4, 64),    // 20
    XT(TILE2(64, 32, 2, 2, 64)),    // 21
    XT(TILE3(32, 64, 2, 2, 64, 2)), // 22: in-workgroup k-split, 8 waves
    XT(TILE3(32, 32, 2, 2, 64, 2)), // 23
    XT(TILE3(32, 32, 2, 2, 32, 4)), // 24: 16 waves
    XT(TILE3(64, 64, 2, 2, 32, 2)), // 25
    TILE3(16, 64, 1, 4, 64, 2), // 26
    XT(TILE3(32, 64, 2, 2, 32, 2)), // 27
    XT(TILE3(16, 64, 1, 4, 32, 4)), // 28
    TILE4(64, 64, 2, 2, 32),    // 29: v3 producer/consumer waves
    XT(TILE4(64, 64, 2, 2, 64)),    // 30
    TILE4(32, 64, 2, 2, 64),    // 31
    XT(TILE4(128, 128, 2, 2, 32)),  // 32
    TILE4(32, 32, 2, 2, 64),    // 33
    XT(TILE4(64, 128, 2, 2, 32)),   // 34
    XT(TILE4(128, 64, 2, 2, 32)),   // 35
    XT(TILE4(32, 128, 2, 2, 64)),   // 36
    XT(TILE5(64, 64, 2, 2, 32, 2)),    // 37: v3, two k-tiles of loads in flight per producer
    XT(TILE5(64, 64, 2, 2, 64, 2)),    // 38
    TILE5(32, 64, 2, 2, 64, 2),    // 39
    XT(TILE5(128, 128, 2, 2, 32, 2)),  // 40
    TILE5(32, 32, 2, 2, 64, 2),    // 41
    XT(TILE5(64, 128, 2, 2, 32, 2)),   // 42
    XT(TILE5(128, 64, 2, 2, 32, 2)),   // 43
    XT(TILE5(32, 128, 2, 2, 64, 2)),   // 44
    XT(TILE5(64, 64, 2, 2, 32, 3)),    // 45: three in flight
    XT(TILE5(64, 64, 2, 2, 64, 3)),    // 46
    XT(TILE5(32, 64, 2, 2, 64, 3)),    // 47
    XT(TILE5(128, 128, 2, 2, 32, 3)),  // 48
    TILE5(32, 32, 2, 2, 64, 3),    // 49
    XT(TILE5(64, 128, 2, 2, 32, 3)),   // 50
    XT(TILE5(128, 64, 2, 2, 32, 3)),   // 51
    XT(TILE5(32, 128, 2, 2, 64, 3)),   // 52
    TILE10(32, 32, 2, 2, 64, 2, 8),   // 53: eight producer waves
    TILE10(32, 32, 2, 2, 64, 3, 8),   // 54
    TILE10(32, 64, 2, 2, 64, 2, 8),   // 55
    TILE10(32, 64, 2, 2, 64, 3, 8),   // 56
    TILE10(64, 64, 2, 2, 64, 2, 8),   // 57
    XT(TILE10(64, 64, 2, 2, 32, 2, 8)),   // 58
    XT(TILE10(32, 128, 2, 2, 64, 2, 8)),  // 59
    XT(TILE10(64, 32, 2, 2, 64, 2, 8)),   // 60
    XT(TILEW(32, 64, 2, 4, 3)),           // 61: weight-streaming consumers (filters in fragment order, usot_conv_pack_wfrag_f32); 1 x 4 waves
    XT(TILEW(32, 64, 2, 4, 2)),           // 62
    XT(TILEW(64, 64, 2, 4, 3)),           // 63
    XT(TILEW2(32, 64, 2, 4, 2)),          // 64: the same, consumer waves 2 x 2 (a wave pair shares its filter fragments through L1)
    XT(TILEW2(64, 64, 2, 4, 2)),          // 65
    XT(TILE11(32, 64, 2, 2, 64, 2, 4)),   // 66: v3 with fragment reads two rounds ahead (PF = 2)
    XT(TILE11(32, 32, 2, 2, 64, 2, 8)),   // 67
    XT(TILES(18, 1)),                     // 68: weight-stationary, K = 2304 (3 x 3 x 256)
    XT(TILES(9, 1)),                      // 69: K = 1152 (3 x 3 x 128)
    XT(TILESB(18, 1)),                    // 70: K = 2304, blocked accumulation (64-product blocks + running total)
    XT(TILES(8, 1)),                      // 71: K = 1024
    XT(TILEP(64, 64, 2, 2, 2, 8)),        // 72: v3 as a persistent stream-K launch (whole-chip rounds)
    XT(TILEP(32, 64, 2, 2, 2, 8)),        // 73
    XT(TILEP(64, 64, 2, 2, 3, 8)),        // 74
    XT(TILEP(64, 64, 2, 2, 2, 4)),        // 75
    XT(TILEP(32, 32, 2, 2, 3, 8)),        // 76
    XT(TILEP32(64, 64, 2, 2, 2, 8)),      // 77: k-tiles of 32
    XT(TILEP32(128, 64, 2, 2, 2, 4)),     // 78
    // (72-78, round 5: parity-green; Conf_Fusion's conv isolated 120 (v3 32 x 64) -> 111 us on tile 74 - 64 x 64 tiles without the
    //  three-round quantisation - but INSIDE the frame 111.6 -> 119.5 us and the graph +13 us; the three search encoders 74.6 -> 71.1
    //  per op, graph +10; shortcut conv + conv1 97 -> 132.  128 x 64 / 64 x 128 / 128 x 128 shapes spill at 768 threads.  Not in
    //  the tuning table)
    // (61-67: parity-green, none faster than v3 - DESIGN.md section 3.1 "what a k-step waits for")
    XT(TILE12(32, 64, 2, 4, 64, 2, 8)),   // 79: v3 with EIGHT consumer waves (two per SIMD) on the same tile: 16 x 16 wave tiles
    XT(TILE12(32, 64, 2, 4, 64, 3, 8)),   // 80
    XT(TILE12(64, 64, 2, 4, 64, 2, 8)),   // 81: 32 x 16 wave tiles
    XT(TILE12(64, 64, 4, 2, 64, 2, 8)),   // 82: 16 x 32
    XT(TILE12(64, 32, 4, 2, 64, 2, 8)),   // 83: 16 x 16
    XT(TILE12(32, 64, 2, 4, 64, 2, 4)),   // 84
    XT(TILE12(64, 64, 2, 4, 64, 3, 8)),   // 85
    XT(TILE13(32, 64, 2, 2, 64, 2, 8)),   // 86: v3 with the consumer's fragment reads hand-scheduled (PF = 3: asm reads two rounds ahead, counted waits)
    XT(TILE13(32, 64, 2, 2, 64, 3, 8)),   // 87
    XT(TILE13(32, 32, 2, 2, 64, 2, 8)),   // 88
    XT(TILE13(32, 32, 2, 2, 64, 3, 8)),   // 89
    XT(TILE13(64, 64, 2, 2, 64, 2, 8)),   // 90
    TILEH(32, 64, 2, 2, 2, 8),        // 91: v3 on SPLIT-fp16 arithmetic (PF = 4: filters pre-split, usot_conv_desc.w_frag = 2 + w_scale)
    TILEH(32, 64, 2, 2, 3, 8),        // 92
    XT(TILEH(32, 64, 2, 2, 4, 8)),        // 93
    TILEH(32, 32, 2, 2, 2, 8),        // 94
    TILEH(32, 32, 2, 2, 3, 8),        // 95
    XT(TILEH(32, 32, 2, 2, 4, 8)),        // 96
    TILEH(64, 64, 2, 2, 2, 8),        // 97
    XT(TILEH(64, 64, 2, 2, 3, 8)),        // 98
    TILEH(32, 64, 2, 2, 2, 4),        // 99
    XT(TILEH(32, 64, 2, 2, 4, 4)),        // 100
    XT(TILEH(64, 64, 2, 2, 4, 8)),        // 101
    XT(TILEH(32, 128, 2, 2, 2, 8)),       // 102
    XT(TILEH(64, 128, 2, 2, 2, 8)),       // 103
    TILEH(32, 32, 2, 2, 2, 4),        // 104
    XT(TILEHD(32, 64, 2, 2, 2, 4)),       // 105: split-fp16 with the filter tile moved by four LDS-DMA wavefronts into five stages (PF = 5)
    TILEHD(32, 64, 2, 2, 3, 4),       // 106
    TILEHD(32, 64, 2, 2, 2, 8),       // 107
    XT(TILEHD(32, 32, 2, 2, 2, 4)),       // 108
    XT(TILEHD(64, 64, 2, 2, 2, 4)),       // 109
    XT(TILEHD(64, 64, 2, 2, 2, 8)),       // 110
    TILEHD(32, 32, 2, 2, 3, 4),       // 111
    XT(TILEHD(64, 64, 2, 2, 3, 4)),       // 112
    XT(TILEHD(32, 64, 2, 2, 4, 4)),       // 113: D = 4 and SIX filter stages (the DMA five k-tiles ahead)
    XT(TILEHD(32, 32, 2, 2, 4, 4)),       // 114
    XT(TILEHD(32, 64, 2, 2, 4, 8)),       // 115
    XT(TILEHX(32, 64, 2, 2)),             // 116: split-fp16, BOTH operands by LDS-DMA (PF = 6): the input map arrives split (usot_conv_desc.x_split), no producer waves
    XT(TILEHX(32, 32, 2, 2)),             // 117
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

int pick_tile(const usot_conv_desc *d, int M)
{
    // Aim for >= ~1 wave of workgroups (256 CUs, up to 4 resident per CU for the small
    // tiles) without shrinking below what the problem needs.
    const int order[] = {1, 2, 4, 5, 7};
    int best = 7;
    for (int id : order) {
        const TileCfg &t = kTiles[id - 1];
        if (t.bn > ((d->Cout + 15) / 16) * 16 && t.bn > 32) continue;
        long blocks = (long)((M + t.bm - 1) / t.bm) * ((d->Cout + t.bn - 1) / t.bn) * d->groups;
        if (blocks >= 512) { best = id; break; }
    }
    if (M <= 16 * 12) best = 8;
    return best;
}

}  // namespace

extern "C" int usot_conv_tile_count(void) { return kNumTiles; }

extern "C" int usot_conv_tile_built(int tile)
{
    if (tile < 1 || tile > kNumTiles) return 0;
    return (kTiles[tile - 1].fn || kTiles[tile - 1].skfn) ? 1 : 0;
}

extern "C" int usot_experiments_built(void)
{
#ifdef USOT_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int usot_conv_tile_info(int tile, int *bm, int *bn)
{
    if (tile < 1 || tile > kNumTiles) return USOT_EINVAL;
    if (bm) *bm = kTiles[tile - 1].bm;
    if (bn) *bn = kTiles[tile - 1].bn;
    return USOT_OK;
}

/* kernel symbol of a tile id as rocprofv3 prints it (for matching bench.py's roofline object
 * with profiles/) */
extern "C" int usot_conv_tile_name(int tile, char *buf, int len)
{
    if (tile < 1 || tile > kNumTiles || !buf || len < 8) return USOT_EINVAL;
    const TileCfg &t = kTiles[tile - 1];
    if (!t.fn && !t.skfn) { snprintf(buf, len, "(tile %d: experiments build only)", tile); return USOT_OK; }
    if (t.nst) { snprintf(buf, len, "conv_wstat_f32<NST=%d,RPS=%d>", t.nst, t.rps); return USOT_OK; }
    if (t.skfn) { snprintf(buf, len, "conv_igemm_f32_v3p<%d,%d,BK=%d,D=%d,NPW=%d>", t.bm, t.bn, t.bk, t.depth, (t.threads - 256) / 64); return USOT_OK; }
    if (t.wfrag == 2 && t.dw == 6) { snprintf(buf, len, "conv_igemm_f32_v3<%d,%d,BK=%d,PF=6>", t.bm, t.bn, t.bk); return USOT_OK; }
    if (t.wfrag == 2 && t.dw == 5) { snprintf(buf, len, "conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=%d,PF=5>", t.bm, t.bn, t.bk, t.depth, (t.threads - 512) / 64); return USOT_OK; }
    if (t.wfrag == 2) { snprintf(buf, len, "conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=%d,PF=4>", t.bm, t.bn, t.bk, t.depth, (t.threads - 256) / 64); return USOT_OK; }
    if (t.wfrag) { snprintf(buf, len, "conv_igemm_f32_ws<%d,%d,D=%d,NPW=%d,DW=%d>", t.bm, t.bn, t.depth, (t.threads - 256) / 64, t.dw); return USOT_OK; }
    if (t.dw == 8 && !t.wfrag) { snprintf(buf, len, "conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=%d,NCW=8>", t.bm, t.bn, t.bk, t.depth, (t.threads - 512) / 64); return USOT_OK; }
    if (t.dw == 3 && !t.wfrag) { snprintf(buf, len, "conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=%d,PF=3>", t.bm, t.bn, t.bk, t.depth, (t.threads - 256) / 64); return USOT_OK; }
    if (t.dw == 2 && !t.wfrag) { snprintf(buf, len, "conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=%d,PF=2>", t.bm, t.bn, t.bk, t.depth, (t.threads - 256) / 64); return USOT_OK; }
    if (t.threads == 768 && t.ksw == 1) { snprintf(buf, len, "conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=8>", t.bm, t.bn, t.bk, t.depth); return USOT_OK; }
    const char *fam = t.threads == 512 && t.ksw == 1 ? "conv_igemm_f32_v3" : (t.stages == 3 ? "conv_igemm_f32_v2" : "conv_igemm_f32");
    if (t.stages == 3 && t.ksw > 1) snprintf(buf, len, "%s<%d,%d,%d,%d> ksw=%d", fam, t.bm, t.bn, t.bk, t.ksw, t.ksw);
    else if (t.depth > 1)           snprintf(buf, len, "%s<%d,%d,BK=%d,D=%d>", fam, t.bm, t.bn, t.bk, t.depth);
    else if (t.stages == 3)         snprintf(buf, len, "%s<%d,%d,BK=%d>", fam, t.bm, t.bn, t.bk);
    else                            snprintf(buf, len, "%s<%d,%d>", fam, t.bm, t.bn);
    return USOT_OK;
}

/* 1 when the tile takes its filters in MFMA fragment order (usot_conv_pack_wfrag_f32, descriptor field w_frag = 1) */
extern "C" int usot_conv_tile_wfrag(int tile)
{
    if (tile < 1 || tile > kNumTiles) return 0;
    return kTiles[tile - 1].wfrag;
}

extern "C" int usot_conv_tile_xsplit(int tile) { return (tile >= 1 && tile <= kNumTiles && kTiles[tile - 1].wfrag == 2 && kTiles[tile - 1].dw == 6) ? 1 : 0; }

/* weight-stationary tiles serve ONE reduction length: K the tile requires (0: any K the other rules allow); their other
 * requirements: Cin % kpanel == 0 (128 or 256, returned through *kpanel), Cout % 32 == 0, ksplit == 1, w_frag == 1 */
extern "C" int usot_conv_tile_kreq(int tile, int *kpanel)
{
    if (tile < 1 || tile > kNumTiles || !kTiles[tile - 1].nst) return 0;
    if (kpanel) *kpanel = 128 * kTiles[tile - 1].rps;
    return kTiles[tile - 1].nst * 128 * kTiles[tile - 1].rps;
}

namespace {
__global__ void pack_wfrag_kernel(const float *__restrict__ w, float *__restrict__ wf, int Cout, int K, long n4)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;        // one float4 of the packed bank
    if (i >= n4) return;
    const int lane = (int)(i & 63);
    const long rest = i >> 6;
    const int r = (int)(rest & 3);
    const long tb = rest >> 2;                                          // cb * KT + kt
    const int KT = K / 64;
    const int kt = (int)(tb % KT), cb = (int)(tb / KT);
    const int row = cb * 16 + (lane & 15), col = kt * 64 + r * 16 + (lane >> 4) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < Cout) v = *(const f32x4 *)(w + (long)row * K + col);
    *(f32x4 *)(wf + i * 4) = v;
}
}  // namespace

/* w [Cout][K] row-major (k = (kh*KW + kw)*Cin + ci, Cin % 64 == 0) -> wf: ceil(Cout/16) blocks of 16 x K floats in the order
 * [block][k-tile of 64][round of 16 k][lane = (k-slot quad)*16 + row][4 consecutive k]: what lane `lane` of a consumer
 * wave feeds v_mfma_f32_16x16x4_f32 as the A operand, one contiguous KiB per wave and round.  Rows past Cout are zero. */
extern "C" int usot_conv_pack_wfrag_f32(void *stream, const float *w, float *wf, int Cout, int K)
{
    if (!w || !wf || Cout < 1 || K < 64 || (K & 63)) return USOT_EINVAL;
    if (((uintptr_t)w | (uintptr_t)wf) & 15) return USOT_EINVAL;
    const long n4 = (long)((Cout + 15) / 16) * 16 * K / 4;
    hipLaunchKernelGGL(pack_wfrag_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wf, Cout, K, n4);
    if (hipGetLastError() != hipSuccess) return USOT_ELAUNCH;
    return USOT_OK;
}

extern "C" int usot_conv_resolve_tile(const usot_conv_desc *d)
{
    if (!d) return USOT_EINVAL;
    if (d->tile != 0) return d->tile;
    return pick_tile(d, d->N * d->OH * d->OW);
}

extern "C" int64_t usot_conv_ws_floats(const usot_conv_desc *d)
{
    if (!d || d->ksplit <= 1) return 0;
    /* slabs + one ticket word per (group, tile) of the smallest tile shape (16 x 32) */
    const int64_t m = (int64_t)d->N * d->OH * d->OW;
    return (int64_t)d->ksplit * d->groups * m * d->Cout + (int64_t)d->groups * ((m + 15) / 16) * ((d->Cout + 31) / 32);
}

namespace {

// validates one descriptor and fills the kernel-side parameter block (tile-independent part)
int fill_params(const usot_conv_desc *d, ConvK &p)
{
    if (!d || !d->x || !d->w || !d->y) return USOT_EINVAL;
    if (d->Cin <= 0 || (d->Cin & 31) || d->Cout <= 0 || d->N <= 0) return USOT_EINVAL;
    if (d->groups < 1 || d->stride < 1 || d->KH < 1 || d->KW < 1) return USOT_EINVAL;
    const int oh = (d->H + 2 * d->pad_h - d->dil_h * (d->KH - 1) - 1) / d->stride + 1;
    const int ow = (d->W + 2 * d->pad_w - d->dil_w * (d->KW - 1) - 1) / d->stride + 1;
    if (oh != d->OH || ow != d->OW || oh <= 0 || ow <= 0) return USOT_EINVAL;
    const int ksplit = d->ksplit > 1 ? d->ksplit : 1;
    if (ksplit > 1 && !d->ws) return USOT_EINVAL;
    // the split-K slabs are addressed through a raw buffer descriptor with 32-bit byte offsets (ws_rsrc): past 2 GiB the
    // offsets would wrap and the accesses fall out of range silently
    if (ksplit > 1 && (int64_t)ksplit * (d->groups > 1 ? d->groups : 1) * d->N * oh * ow * d->Cout * 4 >= 0x7fffffffLL) return USOT_EINVAL;
    p.x = d->x; p.w = d->w; p.bias = d->bias; p.res = d->res; p.y = d->y; p.ws = d->ws;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    p.dil_h = d->dil_h; p.dil_w = d->dil_w;
    p.y_nchw = d->y_nchw;
    p.y_cstride = d->y_cstride > 0 ? d->y_cstride : d->Cout;
    p.y_coff = d->y_coff;
    p.res_cstride = d->res_cstride > 0 ? d->res_cstride : d->Cout;
    p.res_coff = d->res_coff;
    p.act = d->act; p.act2 = d->act2;
    p.act_split = d->act_split > 0 ? d->act_split : (1 << 30);
    p.groups = d->groups;
    p.x_gs = d->x_gs; p.w_gs = d->w_gs; p.b_gs = d->b_gs; p.y_gs = d->y_gs; p.r_gs = d->r_gs;
    p.ksplit = ksplit;
    if (d->defer && ksplit < 2) return USOT_EINVAL;
    p.combine = d->defer ? 0 : 1;            // deferred reduction: the slabs ARE the result (plain stores, no tickets)
    p.P = d->OH * d->OW;
    p.M = d->N * p.P;
    p.K = d->KH * d->KW * d->Cin;
    p.cchunks = d->Cin / 32;
    p.KT = d->KH * d->KW * p.cchunks;
    p.wscale = nullptr; p.zero = nullptr; p.x_split = p.y_split = 0; p.ovf = nullptr;
    p.vec_store = !d->y_nchw && (p.y_cstride % 4 == 0) && (p.y_coff % 4 == 0) &&
                  (!d->res || (p.res_cstride % 4 == 0 && p.res_coff % 4 == 0)) &&
                  ((uintptr_t)d->y % 16 == 0) && (!d->res || (uintptr_t)d->res % 16 == 0) &&
                  (!d->bias || (uintptr_t)d->bias % 16 == 0) &&
                  (d->y_gs % 4 == 0) && (d->r_gs % 4 == 0) && (d->b_gs % 4 == 0);
    if (((uintptr_t)d->x % 16) || ((uintptr_t)d->w % 16) || (d->x_gs % 4) || (d->w_gs % 4))
        return USOT_EINVAL;
    return USOT_OK;
}

}  // namespace

namespace {
// geometry of a persistent stream-K launch of `n` problems on tile tc: fills bt.p[i].MT/NT, sk; returns the workspace floats
// (slabs [tile][part][BM x BN] + four ticket words per tile) or a negative status
int64_t sk_plan(const TileCfg &tc, const usot_conv_desc *d, int n, ConvBatch &bt, SkInfo &sk)
{
    long U = 0, T = 0;
    int ktmax = 1;
    for (int i = 0; i < n; ++i) {
        ConvK &p = bt.p[i];
        if (d[i].Cin % tc.bk || p.ksplit != 1 || d[i].w_frag || d[i].defer) return USOT_EINVAL;
        p.MT = (p.M + tc.bm - 1) / tc.bm;
        p.NT = (d[i].Cout + tc.bn - 1) / tc.bn;
        const int kt = p.K / tc.bk;
        sk.ubase[i] = (int)U;
        sk.tbase[i] = (int)T;
        U += (long)p.groups * p.MT * p.NT * kt;
        T += (long)p.groups * p.MT * p.NT;
        ktmax = kt > ktmax ? kt : ktmax;
    }
    if (U <= 0 || U > 0x3fffffffL) return USOT_EINVAL;
    for (int i = n; i < 5; ++i) { sk.ubase[i] = (int)U; sk.tbase[i] = (int)T; }
    const int usot_dv = usot_device_slot();
    if (usot_dv < 0) return USOT_ESTATE;
    static int slots_d[USOT_MAX_DEV][128] = {};
    int (&slots)[128] = slots_d[usot_dv];                       // resident workgroups of this tile on the current device
    const int tile = (int)(&tc - kTiles);
    if (!slots[tile]) {
        const size_t lds = (size_t)tc.stages * (tc.bm + tc.bn) * (tc.bk + 4) * sizeof(float);
        if (hipFuncSetAttribute((const void *)tc.skfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return USOT_ELAUNCH;
        int dev = 0, cus = 256, occ = 1;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)tc.skfn, tc.threads, lds) != hipSuccess || occ < 1) occ = 1;
        slots[tile] = cus * (occ > 2 ? 2 : occ);
    }
    long G = slots[tile];
    if (G > U / 4) G = U / 4 > 0 ? U / 4 : 1;          // at least four k-tiles per workgroup
    sk.G = (int)G;
    const long share = U / G;                          // the smallest share
    sk.pmax = (int)((ktmax + share - 1) / share) + 1;  // a tile of KT units meets at most ceil(KT / share) + 1 shares
    const int64_t fl = (int64_t)T * sk.pmax * tc.bm * tc.bn + (int64_t)T * 4;
    if ((int64_t)T * sk.pmax * tc.bm * tc.bn * 4 >= 0x7fffffffLL) return USOT_EINVAL;
    return fl;
}
}  // namespace

/* workspace floats a persistent stream-K tile (usot_conv_tile_streamk(tile) == 1) needs for this batch of problems: the partial
 * tiles of the workgroup shares that end inside a tile + four ticket words per tile.  d[0].ws must hold that many floats, ZERO
 * before the first launch (the kernel leaves the tickets zero); 0 for the other tiles, negative = invalid arguments. */
extern "C" int64_t usot_conv_streamk_ws_floats(const usot_conv_desc *d, int n, int tile)
{
    if (!d || n < 1 || n > 4 || tile < 1 || tile > kNumTiles) return USOT_EINVAL;
    const TileCfg &tc = kTiles[tile - 1];
    if (!tc.skfn) return 0;
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    ConvBatch bt;
    SkInfo sk;
    bt.n = n;
    for (int i = 0; i < n; ++i) {
        usot_conv_desc c = d[i];
        if (!c.y) c.y = (float *)c.x;                  // sizing only
        int rc = fill_params(&c, bt.p[i]);
        if (rc != USOT_OK) return rc;
    }
    return sk_plan(tc, d, n, bt, sk);
}

extern "C" int usot_conv_tile_streamk(int tile) { return (tile >= 1 && tile <= kNumTiles && kTiles[tile - 1].skfn) ? 1 : 0; }

extern "C" int usot_conv2d_batch_f32(void *stream, const usot_conv_desc *d, int n)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    if (!d || n < 1 || n > 4) return USOT_EINVAL;
    ConvBatch bt;
    bt.n = n;
    int rc;
    for (int i = 0; i < n; ++i)
        if ((rc = fill_params(&d[i], bt.p[i])) != USOT_OK) return rc;
    int tile = d[0].tile;
    if (tile == 0) {                       // heuristic on the largest problem
        int big = 0;
        for (int i = 1; i < n; ++i)
            if ((long)bt.p[i].M * bt.p[i].Cout * bt.p[i].K > (long)bt.p[big].M * bt.p[big].Cout * bt.p[big].K) big = i;
        tile = pick_tile(&d[big], bt.p[big].M);
    }
    if (tile < 1 || tile > kNumTiles) return USOT_EINVAL;
    const TileCfg &tc = kTiles[tile - 1];
    if (!tc.fn && !tc.skfn) return USOT_ENOTBUILT;
    long blocks = 0;
    if (tc.skfn) {                         // persistent stream-K: a resident set of workgroups shares the (tile, k-tile) units
        SkInfo sk;
        const int64_t need = sk_plan(tc, d, n, bt, sk);
        if (need < 0) return (int)need;
        if (!d[0].ws) return USOT_EINVAL;
        for (int i = 0; i < n; ++i) bt.p[i].ws = d[0].ws;         // one workspace for the batch
        for (int i = 0; i < 5; ++i) bt.start[i] = 0;
        const size_t lds = (size_t)tc.stages * (tc.bm + tc.bn) * (tc.bk + 4) * sizeof(float);
        hipLaunchKernelGGL(tc.skfn, dim3((unsigned)sk.G), dim3(tc.threads), lds, (hipStream_t)stream, bt, sk);
        if (hipGetLastError() != hipSuccess) return USOT_ELAUNCH;
        return USOT_OK;
    }
    if (tc.nst) {                          // weight-stationary tiles: one workgroup per (group, 32 channels, pixel range)
        double work = 0;
        for (int i = 0; i < n; ++i) work += (double)bt.p[i].M * bt.p[i].Cout * bt.p[i].groups;
        int cus = 256;
        {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
        }
        for (int i = 0; i < n; ++i) {
            ConvK &p = bt.p[i];
            const int pk = 128 * tc.rps;
            if (p.K != tc.nst * pk || (d[i].Cin % pk) || (d[i].Cout & 31) || p.ksplit != 1 || d[i].w_frag != 1) return USOT_EINVAL;
            if (p.groups > 1 && (p.w_gs % ((long)16 * p.K))) return USOT_EINVAL;
            if ((long)p.N * p.H * p.W * p.Cin >= (1L << 31)) return USOT_EINVAL;        // 32-bit element offsets in the loader
            p.MT = (p.M + 31) / 32;
            p.NT = d[i].Cout / 32;
            // pixel ranges: this problem's share of the CUs (one workgroup per CU: 8 waves x ~220 VGPRs), by its share of the work
            const double share = (double)p.M * p.Cout * p.groups / work;
            int pr = (int)(cus * share / ((double)p.groups * p.NT));
            if (d[i].ksplit < 0) pr = -d[i].ksplit;            // experiments: ksplit = -PR forces the number of pixel ranges
            p.pr = pr < 1 ? 1 : (pr > p.MT ? p.MT : pr);
            bt.start[i] = (int)blocks;
            blocks += (long)p.groups * p.NT * p.pr;
        }
        for (int i = n; i < 5; ++i) bt.start[i] = (int)blocks;
        const size_t lds = (size_t)4 * 32 * (128 * tc.rps + 4) * sizeof(float) + 8 * 4 * 64 * 16;
        static bool raised_d[USOT_MAX_DEV][128] = {};
    bool (&raised)[128] = raised_d[usot_dv];
        if (!raised[tile]) {
            if (hipFuncSetAttribute((const void *)tc.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return USOT_ELAUNCH;
            raised[tile] = true;
        }
        hipLaunchKernelGGL(tc.fn, dim3((unsigned)blocks), dim3(tc.threads), lds, (hipStream_t)stream, bt);
        if (hipGetLastError() != hipSuccess) return USOT_ELAUNCH;
        return USOT_OK;
    }
    for (int i = 0; i < n; ++i) {
        ConvK &p = bt.p[i];
        if (d[i].Cin % tc.bk) return USOT_EINVAL;
        if (p.ksplit > p.K / tc.bk) return USOT_EINVAL;
        // filters in fragment order iff the tile streams them (a row-major bank under a streaming tile, or the reverse,
        // would compute garbage silently); group strides must keep whole 16-row blocks
        if (d[i].w_frag != tc.wfrag) return USOT_EINVAL;
        if (tc.wfrag == 2 && (!d[i].w_scale || ((uintptr_t)d[i].w_scale & 3))) return USOT_EINVAL;
        p.wscale = d[i].w_scale;
        if ((uintptr_t)d[i].ovf & 3) return USOT_EINVAL;
        p.ovf = tc.wfrag == 2 ? d[i].ovf : nullptr;
        // split maps: only the split-fp16 tiles write them (vectorised NHWC epilogue, whole 64-channel blocks, no split-K), only the
        // all-DMA tiles read them - and those read nothing else
        p.x_split = d[i].x_split ? 1 : 0;
        p.y_split = d[i].y_split ? 1 : 0;
        if (p.x_split != (tc.wfrag == 2 && tc.dw == 6 ? 1 : 0)) return USOT_EINVAL;
        if (p.y_split && (tc.wfrag != 2 || !p.vec_store || (d[i].Cout & 63) || (p.y_coff & 63) || (p.y_cstride & 63) || p.ksplit > 1 || d[i].res)) return USOT_EINVAL;
        if (p.x_split) {
            static const float *zero_page_d[USOT_MAX_DEV] = {};
    const float *&zero_page = zero_page_d[usot_dv];
            if (!zero_page) {
                void *zp = nullptr;
                if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero16_f32)) != hipSuccess || !zp) return USOT_ELAUNCH;
                zero_page = (const float *)zp;
            }
            p.zero = zero_page;
        }
        if (tc.wfrag == 1 && p.groups > 1 && (p.w_gs % ((long)16 * p.K))) return USOT_EINVAL;
        p.MT = (p.M + tc.bm - 1) / tc.bm;
        p.NT = (d[i].Cout + tc.bn - 1) / tc.bn;
        bt.start[i] = (int)blocks;
        blocks += (long)p.MT * p.NT * p.groups * p.ksplit;
    }
    for (int i = n; i < 5; ++i) bt.start[i] = (int)blocks;
    if (blocks <= 0 || blocks > 0x7fffffffL) return USOT_EINVAL;
    // the weight-streaming tiles stage the activation operand only
    // v3 / ws with BK = 64: unpadded swizzled rows; every other form pads a row by 4 floats
    const bool v3fam = tc.stages == 3 && tc.ksw == 1 && tc.threads >= 512;
#ifdef USOT_V3_SWZ
    const int ld = (tc.bk == 64 && (v3fam || tc.wfrag)) ? 64 : tc.bk + 4;
#else
    const int ld = (tc.bk == 64 && tc.wfrag == 1) ? 64 : tc.bk + 4;
    (void)v3fam;
#endif
    size_t lds = (size_t)tc.ksw * tc.stages * (tc.bm + (tc.wfrag == 1 ? 0 : tc.bn)) * ld * sizeof(float);
    if (tc.wfrag == 2 && tc.dw == 5) lds = (size_t)(3 * tc.bm + (tc.depth >= 4 ? 6 : 5) * tc.bn) * ld * sizeof(float);      // three activation + five (six) filter stages
    if (tc.wfrag == 2 && tc.dw == 6) lds = (size_t)(6 * tc.bm + 6 * tc.bn) * ld * sizeof(float);                           // six of each
    if (lds > 64 * 1024) {
        static bool raised_d[USOT_MAX_DEV][128] = {};
    bool (&raised)[128] = raised_d[usot_dv];
        if (!raised[tile]) {
            if (hipFuncSetAttribute((const void *)tc.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return USOT_ELAUNCH;
            raised[tile] = true;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tc.fn, dim3((unsigned)blocks), dim3(tc.threads), lds, s, bt);
    if (hipGetLastError() != hipSuccess) return USOT_ELAUNCH;
    return USOT_OK;
}

extern "C" int usot_conv2d_f32(void *stream, const usot_conv_desc *d)
{
    return usot_conv2d_batch_f32(stream, d, 1);
}
