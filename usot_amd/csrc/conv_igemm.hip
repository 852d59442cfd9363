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
template <int BM, int BN, int WM, int WN, int BK, int D = 1, int NPW = 4>
__global__ __launch_bounds__(256 + 64 * NPW) void conv_igemm_f32_v3(const ConvBatch bt)
{
    int pi = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < bt.n && (int)blockIdx.x >= bt.start[q]) pi = q;
    const ConvK &p = bt.p[pi];
    const int bid0 = (int)blockIdx.x - bt.start[pi];
    static_assert(WM * WN == 4, "4 consumer wavefronts");
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int LD = BK + 4;
    constexpr int CPR = BK / 4;
    constexpr int NPT = 64 * NPW;                 // producer threads
    constexpr int RPP = NPT / CPR;
    constexpr int XI = (BM + RPP - 1) / RPP, WI = (BN + RPP - 1) / RPP;
    constexpr int NR = BK / 16;
    constexpr int STAGE = (BM + BN) * LD;

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
#pragma unroll
            for (int i = 0; i < WI; ++i) {
#ifdef USOT_ABL_NOLOAD
                wr[d][i] = f32x4{1.f, 1.f, 1.f, 1.f};
#else
                wr[d][i] = *(const f32x4 *)wp[i];
#endif
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
#ifdef USOT_ABL_NOSTORE
            return;
#endif
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
        auto step = [&](auto dc, int t) {
            USOT_STAMP(4, t);
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
    const int fx_off = (wm * TM * 16 + l15) * LD + quad * 4;
    const int fw_off = BM * LD + (wn * TN * 16 + l15) * LD + quad * 4;
    f32x4 fw[2][TN], fx[2][TM];
    auto read_frags = [&](int st, int r, int slot) {
#ifdef USOT_ABL_NOREAD
        return;
#endif
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
#ifdef USOT_TRACE
    unsigned *trc = (p.ksplit == 1 && p.ws && bid0 == 0 && tid == 0) ? (unsigned *)p.ws : nullptr;
#endif
    if (nt > 0) read_frags(0, 0, 0);
    int st = 0;
    for (int t = 0; t < nt; ++t) {
        const int st1 = st == 2 ? 0 : st + 1;
        USOT_STAMP(0, t);
        flush();                                      // the previous k-tile's block (zeros at t = 0)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r + 1 < NR) read_frags(st, r + 1, (r + 1) & 1);
            else if (t + 1 < nt) read_frags(st1, 0, 0);
            mma(r & 1, r == 0);
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

struct TileCfg { int bm, bn, bk, stages, ksw; void (*fn)(const ConvBatch); int threads; int depth; };

#define TILE(bm, bn, wm, wn) { bm, bn, 32, 2, 1, conv_igemm_f32<bm, bn, wm, wn>, 256, 1 }
#define TILE2(bm, bn, wm, wn, bk) { bm, bn, bk, 3, 1, conv_igemm_f32_v2<bm, bn, wm, wn, bk>, 256, 1 }
#define TILE3(bm, bn, wm, wn, bk, ksw) { bm, bn, bk, 3, ksw, conv_igemm_f32_v2<bm, bn, wm, wn, bk, ksw>, 256 * ksw, 1 }
#define TILE4(bm, bn, wm, wn, bk) { bm, bn, bk, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, bk>, 512, 1 }
#define TILE10(bm, bn, wm, wn, bk, d, npw) { bm, bn, bk, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, bk, d, npw>, 256 + 64 * npw, d }
#define TILE5(bm, bn, wm, wn, bk, d) { bm, bn, bk, 3, 1, conv_igemm_f32_v3<bm, bn, wm, wn, bk, d>, 512, d }
const TileCfg kTiles[] = {
    TILE(128, 128, 2, 2),   // 1: batched backbone
    TILE(128, 64, 2, 2),    // 2
    TILE(64, 128, 2, 2),    // 3
    TILE(64, 64, 2, 2),     // 4
    TILE(32, 64, 2, 2),     // 5
    TILE(64, 32, 2, 2),     // 6
    TILE(32, 32, 2, 2),     // 7
    TILE(16, 64, 1, 4),     // 8: tiny M (template-side encoders)
    TILE(16, 128, 1, 4),    // 9
    TILE(32, 128, 2, 2),    // 10
    TILE2(128, 128, 2, 2, 32),  // 11: v2 (3-stage, mid-tile barrier)
    TILE2(64, 64, 2, 2, 32),    // 12
    TILE2(64, 64, 2, 2, 64),    // 13
    TILE2(32, 64, 2, 2, 32),    // 14
    TILE2(32, 64, 2, 2, 64),    // 15
    TILE2(32, 32, 2, 2, 64),    // 16
    TILE2(64, 128, 2, 2, 32),   // 17
    TILE2(128, 64, 2, 2, 32),   // 18
    TILE2(32, 128, 2, 2, 32),   // 19
    TILE2(16, 64, 1, 4, 64),    // 20
    TILE2(64, 32, 2, 2, 64),    // 21
    TILE3(32, 64, 2, 2, 64, 2), // 22: in-workgroup k-split, 8 waves
    TILE3(32, 32, 2, 2, 64, 2), // 23
    TILE3(32, 32, 2, 2, 32, 4), // 24: 16 waves
    TILE3(64, 64, 2, 2, 32, 2), // 25
    TILE3(16, 64, 1, 4, 64, 2), // 26
    TILE3(32, 64, 2, 2, 32, 2), // 27
    TILE3(16, 64, 1, 4, 32, 4), // 28
    TILE4(64, 64, 2, 2, 32),    // 29: v3 producer/consumer waves
    TILE4(64, 64, 2, 2, 64),    // 30
    TILE4(32, 64, 2, 2, 64),    // 31
    TILE4(128, 128, 2, 2, 32),  // 32
    TILE4(32, 32, 2, 2, 64),    // 33
    TILE4(64, 128, 2, 2, 32),   // 34
    TILE4(128, 64, 2, 2, 32),   // 35
    TILE4(32, 128, 2, 2, 64),   // 36
    TILE5(64, 64, 2, 2, 32, 2),    // 37: v3, two k-tiles of loads in flight per producer
    TILE5(64, 64, 2, 2, 64, 2),    // 38
    TILE5(32, 64, 2, 2, 64, 2),    // 39
    TILE5(128, 128, 2, 2, 32, 2),  // 40
    TILE5(32, 32, 2, 2, 64, 2),    // 41
    TILE5(64, 128, 2, 2, 32, 2),   // 42
    TILE5(128, 64, 2, 2, 32, 2),   // 43
    TILE5(32, 128, 2, 2, 64, 2),   // 44
    TILE5(64, 64, 2, 2, 32, 3),    // 45: three in flight
    TILE5(64, 64, 2, 2, 64, 3),    // 46
    TILE5(32, 64, 2, 2, 64, 3),    // 47
    TILE5(128, 128, 2, 2, 32, 3),  // 48
    TILE5(32, 32, 2, 2, 64, 3),    // 49
    TILE5(64, 128, 2, 2, 32, 3),   // 50
    TILE5(128, 64, 2, 2, 32, 3),   // 51
    TILE5(32, 128, 2, 2, 64, 3),   // 52
    TILE10(32, 32, 2, 2, 64, 2, 8),   // 53: eight producer waves
    TILE10(32, 32, 2, 2, 64, 3, 8),   // 54
    TILE10(32, 64, 2, 2, 64, 2, 8),   // 55
    TILE10(32, 64, 2, 2, 64, 3, 8),   // 56
    TILE10(64, 64, 2, 2, 64, 2, 8),   // 57
    TILE10(64, 64, 2, 2, 32, 2, 8),   // 58
    TILE10(32, 128, 2, 2, 64, 2, 8),  // 59
    TILE10(64, 32, 2, 2, 64, 2, 8),   // 60
    // (D = 4 and 6 variants of 53-60 were tried inside the frame graph, scripts/tune_frame.py: no layer got faster —
    //  e.g. layer3's 3x3 908 us/frame at D = 2, 909 at D = 4 and 6 — so a batch-1 k-step is not waiting for loads)
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
    if (t.threads == 768 && t.ksw == 1) { snprintf(buf, len, "conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=8>", t.bm, t.bn, t.bk, t.depth); return USOT_OK; }
    const char *fam = t.threads == 512 && t.ksw == 1 ? "conv_igemm_f32_v3" : (t.stages == 3 ? "conv_igemm_f32_v2" : "conv_igemm_f32");
    if (t.stages == 3 && t.ksw > 1) snprintf(buf, len, "%s<%d,%d,%d,%d> ksw=%d", fam, t.bm, t.bn, t.bk, t.ksw, t.ksw);
    else if (t.depth > 1)           snprintf(buf, len, "%s<%d,%d,BK=%d,D=%d>", fam, t.bm, t.bn, t.bk, t.depth);
    else if (t.stages == 3)         snprintf(buf, len, "%s<%d,%d,BK=%d>", fam, t.bm, t.bn, t.bk);
    else                            snprintf(buf, len, "%s<%d,%d>", fam, t.bm, t.bn);
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
    p.combine = 1;
    p.P = d->OH * d->OW;
    p.M = d->N * p.P;
    p.K = d->KH * d->KW * d->Cin;
    p.cchunks = d->Cin / 32;
    p.KT = d->KH * d->KW * p.cchunks;
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

extern "C" int usot_conv2d_batch_f32(void *stream, const usot_conv_desc *d, int n)
{
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
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        ConvK &p = bt.p[i];
        if (d[i].Cin % tc.bk) return USOT_EINVAL;
        if (p.ksplit > p.K / tc.bk) return USOT_EINVAL;
        p.MT = (p.M + tc.bm - 1) / tc.bm;
        p.NT = (d[i].Cout + tc.bn - 1) / tc.bn;
        bt.start[i] = (int)blocks;
        blocks += (long)p.MT * p.NT * p.groups * p.ksplit;
    }
    for (int i = n; i < 5; ++i) bt.start[i] = (int)blocks;
    if (blocks <= 0 || blocks > 0x7fffffffL) return USOT_EINVAL;
    const size_t lds = (size_t)tc.ksw * tc.stages * (tc.bm + tc.bn) * (tc.bk + 4) * sizeof(float);
    if (lds > 64 * 1024) {
        static bool raised[128] = {false};
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
