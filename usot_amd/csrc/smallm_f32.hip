// Small-M fp32 kernels of the batch-1 frame.  At batch 1 a backbone layer has M = 3969 or 961 pixels and 0.1-0.5 GFLOP:
// on the tiled implicit-GEMM kernels (conv_igemm.hip) such a layer is a launch, one cold round trip for the map the
// previous kernel wrote, a handful of k-steps and a boundary - 9-14 us for ~1 us of matrix work.  The kernels here share
// one form instead: a workgroup owns 16 PIXELS (the MFMA's B operand, kept whole-K in LDS), the filters are pre-packed
// in MFMA-fragment order and streamed from L2 straight into registers (no LDS staging of filters, no k-tile barriers),
// and consecutive layers that share their pixel tile run in ONE launch with the intermediate map in LDS.
//   pw_pair_f32_kernel        conv3 + residual + ReLU  ->  next conv1                      (layer2; optional channel slices)
//   pw_triple_f32_kernel      conv2 (3x3)  ->  conv3 + residual + ReLU  ->  next conv1     (layer1)
//   pw_single_f32_kernel      one 1x1 conv (+ residual)                                    (layer3's expansion convs)
//   stream_conv3x3_f32_kernel one 3x3 / stride-1 conv                                      (layer2's conv2)
// DESIGN.md section 3.1 has the measurements (and the shapes where the tiled kernels stay faster).
//
// Fused pair of pointwise convolutions for the fp32 batch-1 frame: a bottleneck's conv3 + BN + residual + ReLU
// (modules.py:48-56) and the NEXT block's conv1 + BN + ReLU (modules.py:40-42; or the neck's 1x1, connect.py:294-300)
// in one launch.
//
// Why: at batch 1 the backbone's 1x1 convolutions are 0.1-0.5 GFLOP each and cost 9-14 us per launch — launch, one
// cold round trip for the activations the previous kernel wrote, a handful of k-steps, the boundary — i.e. 16-36
// TFLOP/s (per-op spans of the frame: layer1 + layer2 29 % of the frame, layer3's 1x1 layers another 18 %).  The pair
// shares its pixel tile: Y never leaves the CU between the two GEMMs, one launch and one cold round trip disappear.
//
//   Y[M][CO] = relu(T2[M][CM] . W3^T + b3 + R[M][CO])        (stored: it is the next block's residual)
//   T [M][CN] = act2(Y . W1^T + b1)
//
// One workgroup = 16 pixels, eight wavefronts.  v_mfma_f32_16x16x4_f32 with the conv kernel's operand roles (filters =
// A, pixels = B: a lane's accumulator is 4 consecutive channels of one pixel, 16-byte epilogue accesses) and its
// ds_read_b128 trick (a lane reads 4 consecutive k and issues 4 MFMAs; quad q owns k-slot q).  Both filter banks are
// pre-packed in FRAGMENT order (one contiguous KiB per 16 channels x 16 k, see usot_pw_pair_f32 in usot_hip.h) and
// stream straight from L2 into a ring of eight register fragments per wave; only the pixel tile and Y go through LDS.
// When GEMM2 has fewer column blocks than waves its k range is split over wave groups, meeting in LDS (fixed order).
//
// S > 1 — few pixel tiles (M = 961: 61): S workgroups share a pixel tile, each owning CO / S channels of Y: GEMM1 for its
// slice, then the partial product of GEMM2 over that slice of k.  They meet through a workspace exactly as the conv
// kernels' in-launch split-K does: write-through slab stores, one relaxed ticket per pixel tile, the last arriver sums
// the S slabs in slice order and applies bias / activation (no fences; tickets are zero before the first launch and
// reset by the last arriver).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

#ifndef USOT_RING          // filter fragments (1 KiB per wave each) a wave keeps in flight
#define USOT_RING 8
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

struct PwF {
    const float *t2, *w3p, *b3, *res, *w1p, *b1;
    float *y, *t, *ws;
    int M, act2;
    int parts = 0;                 // > 1: t2 = `parts` partial sums [parts][M][CM] of the producing conv (usot_conv_desc.defer)
    const float *tbias = nullptr;  //      ... staged as relu(sum in part order + tbias)
    int rparts = 0;                // > 1: res = `rparts` partial sums [rparts][M][CO] of the shortcut conv; residual = sum + rbias
    const float *rbias = nullptr;
    int *ovf = nullptr;            // split-fp16 form: sticky "a sum was not finite" word (usot_pw_pair_desc.ovf)
};

// acc[u] (u < CBW) = sum over rounds [r0, r0 + RS) of W fragment (cb0 + u, r) x the B operand rows in LDS.
// wf: the bank's fragment (cb, r) at wf[(cb * RT + r) * 64] (lane offset already applied); bs: this lane's LDS row.
// A ring of PF filter fragments (1 KiB per wave each) stays in flight; gemm_prefetch fills it (callable before the B
// operand is ready: in front of the barrier that publishes it), gemm_run consumes and refills it.
// BLO > 0: SPLIT-fp16 operands (conv_igemm.hip, PF = 4: every value as hi + lo fp16 of value x a power of two, a product block
// as w_lo x_hi + w_hi x_lo + w_hi x_hi on v_mfma_f32_16x16x32_f16, fp32 accumulation).  The bank then holds, per column block
// and 32-k step, a hi fragment followed by a lo fragment (1 KiB each: lane (quad, row) = 8 halves of k = 32 step + 8 quad ..),
// i.e. the same RT fragments per column block; a B row in LDS is its K hi halves followed by its K lo halves (BLO = K / 2
// floats in), so lane (l15, quad) finds the hi fragment of step s at float 16 s + 4 quad as before and the lo one BLO further.
template <int CBW, int RS, int RT, int PF, int BLO = 0>
struct GemmRing {
    static constexpr int N = CBW * RS;
    static_assert(BLO == 0 || (RS % 4 == 0 && PF % 2 == 0), "split-fp16: whole 64-k blocks, fragments in pairs");
    static_assert(PF <= N, "ring deeper than the work");
    f32x4 ring[PF];
    __device__ __forceinline__ f32x4 frag(const f32x4 *__restrict__ wf, int cb0, int r0, int n) const
    {
        return wf[((cb0 + n / RS) * RT + r0 + n % RS) * 64];
    }
    __device__ __forceinline__ void prefetch(const f32x4 *__restrict__ wf, int cb0, int r0)
    {
#pragma unroll
        for (int n = 0; n < PF; ++n) ring[n] = frag(wf, cb0, r0, n);
    }
    __device__ __forceinline__ void run(const f32x4 *__restrict__ wf, const float *bs, int cb0, int r0, f32x4 (&acc)[CBW])
    {
        // Blocked accumulation (conv_igemm.hip, blocked_mma): four rounds = 64 products go into one of two alternating block
        // accumulators (first MFMA with C = 0); a block joins the running total when its register is due for reuse, two
        // blocks = 32 MFMAs later, so no add ever waits for the matrix pipe; the total is a BlockTotal (common.h).  Everything below is unrolled: static registers.
        constexpr int G = 4;                              // rounds per block
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 part[CBW][2];
        BlockTotal tot[CBW];                              // running total of finished blocks (common.h)
#pragma unroll
        for (int u = 0; u < CBW; ++u) { part[u][0] = part[u][1] = zero; tot[u].clear(); }
        if constexpr (BLO > 0) {
            auto h = [](const f32x4 &v) { return __builtin_bit_cast(f16x8_t, v); };
#pragma unroll
            for (int n = 0; n < N; n += 2) {
                const int step = (r0 + n % RS) >> 1;
                const f32x4 bh = *(const f32x4 *)(bs + step * 16), bl = *(const f32x4 *)(bs + BLO + step * 16);
                const f32x4 ah = ring[n % PF], al = ring[(n + 1) % PF];
                if (n + PF < N) ring[n % PF] = frag(wf, cb0, r0, n + PF);
                if (n + 1 + PF < N) ring[(n + 1) % PF] = frag(wf, cb0, r0, n + 1 + PF);
                const int u = n / RS, rr = n % RS, g = rr / G, s = g & 1;
                const bool first = rr % G == 0;
                if (first && g >= 2) tot[u].add(part[u][s]);
                part[u][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(al), h(bh), first ? zero : part[u][s], 0, 0, 0);
                part[u][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(ah), h(bl), part[u][s], 0, 0, 0);
                part[u][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h(ah), h(bh), part[u][s], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const f32x4 b = *(const f32x4 *)(bs + (r0 + n % RS) * 16);
            const f32x4 a = ring[n % PF];
            if (n + PF < N) ring[n % PF] = frag(wf, cb0, r0, n + PF);
            const int u = n / RS, rr = n % RS, g = rr / G, s = g & 1;
            const bool first = rr % G == 0;
            if (first && g >= 2) tot[u].add(part[u][s]);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                part[u][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[c], (first && c == 0) ? zero : part[u][s], 0, 0, 0);
        }
        }
#pragma unroll
        for (int u = 0; u < CBW; ++u) {                   // the last one or two blocks of every column block are still open
            constexpr int GL = (RS - 1) / G;              // index of the last block
            if (GL >= 1) tot[u].add(part[u][(GL & 1) ^ 1]);
            tot[u].add(part[u][GL & 1]);
            acc[u] = tot[u].get();
        }
    }
};

template <int CBW, int RS, int RT, int BLO = 0>
__device__ __forceinline__ void gemm_blocks(const f32x4 *__restrict__ wf, const float *bs, int cb0, int r0, f32x4 (&acc)[CBW])
{
    GemmRing<CBW, RS, RT, (CBW * RS < USOT_RING ? CBW * RS : USOT_RING), BLO> g;
    g.prefetch(wf, cb0, r0);
    g.run(wf, bs, cb0, r0, acc);
}

// four consecutive values of an LDS operand row in the split-fp16 layout: halves k .. k + 3 of the hi plane and of the lo plane
// (`row` = the row's first byte, `plane` = bytes from the hi plane to the lo plane = 2 K); the values are scaled by 8 first
__device__ __forceinline__ void store_split4(char *row, int k, int plane, f32x4 v)
{
    v *= 8.0f;
    const uint32_t hi0 = usot_pack2_lp<true>(v[0], v[1]), hi1 = usot_pack2_lp<true>(v[2], v[3]);
    const usot_f16x2 h0 = __builtin_bit_cast(usot_f16x2, hi0), h1 = __builtin_bit_cast(usot_f16x2, hi1);
    const uint32_t lo0 = usot_pack2_lp<true>(v[0] - (float)h0[0], v[1] - (float)h0[1]);
    const uint32_t lo1 = usot_pack2_lp<true>(v[2] - (float)h1[0], v[3] - (float)h1[1]);
    *(u32x2_t *)(row + k * 2) = u32x2_t{hi0, hi1};
    *(u32x2_t *)(row + plane + k * 2) = u32x2_t{lo0, lo1};
}

// split-fp16 range contract (conv_igemm.hip, usot_conv_desc.ovf): a value beyond the fp16 window was staged as hi = inf, lo = -inf and
// every sum it entered is NaN; report it in the caller's sticky word before an activation makes it finite again
__device__ __forceinline__ void note_not_finite(int *ovf, const f32x4 &a)
{
    if (!ovf) return;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) bad |= !(__builtin_fabsf(a[e]) <= 3.4028234664e38f);
    if (bad) __hip_atomic_store(ovf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Everything after "the pixel tile is in LDS": GEMM1 + residual + ReLU -> Y (global + LDS), GEMM2 (+ wave-group / workgroup
// meeting) -> T.  Xs: the [16][CM + 4] tile (published by a barrier before the call); Ys / Ps: scratch.
// H16: split-fp16 operands (GemmRing, BLO > 0): both banks pre-split with their per-row factors 1 / (row scale x 8) appended
// (w3p[COT CM ..], w1p[CN COT ..]: hip.pw_pair_s16_pack), the pixel tile staged split by the caller, the Y tile split here.
template <int CM, int COT, int CN, int S, bool H16 = false>
__device__ __forceinline__ void pair_tail(const PwF &p, const float *Xs, float *Ys, float *Ps, int pt, int sl)
{
    constexpr int NW = 8, BM = 16;
    constexpr int CO = COT / S;                           // channels of Y this workgroup owns
    constexpr int R1 = CM / 16, NB1 = CO / 16;            // GEMM1: rounds of 16 k, column blocks of 16 channels
    constexpr int R2 = CO / 16, NB2 = CN / 16, R2T = COT / 16;
    constexpr int CBW1 = NB1 / NW;
    constexpr int KS = NB2 >= NW ? 1 : NW / NB2;          // k-slices of GEMM2 over wave groups
    constexpr int CBW2 = NB2 >= NW ? NB2 / NW : 1;        // column blocks per wave in GEMM2
    constexpr int RS2 = R2 / KS;
    static_assert(NB1 % NW == 0 && (NB2 % NW == 0 || NW % NB2 == 0) && R2 % KS == 0, "shape");
    static_assert(S == 1 || KS == 1, "the sliced form keeps GEMM2's k range of a workgroup in one piece");
    constexpr int XP = CM + 4, YP = CO + 4, PP = CN + 4;  // LDS row pitches (floats): 16 B aligned, rows spread over banks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, quad = lane >> 4;
    const int m = pt * BM + l15;
    const bool mok = m < p.M;

    // ---- GEMM1: K = CM from the pixel tile; this wave's CBW1 column blocks of the slice (residual and bias fly under it)
    {
        f32x4 rr[CBW1], bb[CBW1], sc[H16 ? CBW1 : 1];
#pragma unroll
        for (int u = 0; u < CBW1; ++u) {
            const int cog = sl * CO + (wave * CBW1 + u) * 16 + quad * 4;
            if constexpr (H16) sc[u] = *(const f32x4 *)(p.w3p + (long)COT * CM + cog);
            rr[u] = mok ? *(const f32x4 *)(p.res + (long)m * COT + cog) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.rparts > 1 && mok) {       // deferred split-K reduction of the shortcut convolution: parts in order, then its bias
                for (int q = 1; q < p.rparts; ++q) rr[u] += *(const f32x4 *)(p.res + ((long)q * p.M + m) * COT + cog);
                rr[u] += *(const f32x4 *)(p.rbias + cog);
            }
            bb[u] = *(const f32x4 *)(p.b3 + cog);
        }
        f32x4 acc[CBW1];
        gemm_blocks<CBW1, R1, R1, H16 ? CM / 2 : 0>((const f32x4 *)p.w3p + lane + (long)sl * NB1 * R1 * 64, Xs + l15 * XP + quad * 4, wave * CBW1, 0, acc);
#pragma unroll
        for (int u = 0; u < CBW1; ++u) {
            const int co = (wave * CBW1 + u) * 16 + quad * 4;              // within the slice
            if constexpr (H16) {
                acc[u] *= sc[u];
                note_not_finite(p.ovf, acc[u]);       // before the ReLU below turns an overflow's NaN into a finite 0
            }
            f32x4 v = acc[u] + bb[u] + rr[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            if (mok) *(f32x4 *)(p.y + (long)m * COT + sl * CO + co) = v;
            if constexpr (H16) store_split4((char *)(Ys + l15 * YP), co, CO * 2, v);
            else *(f32x4 *)(Ys + l15 * YP + co) = v;
        }
    }
    __syncthreads();                                      // Y tile (slice) complete

    // ---- GEMM2: K = CO from LDS; wave -> (column blocks, k-slice)
    const int ksl = KS > 1 ? wave / NB2 : 0;
    const int cb0 = KS > 1 ? wave % NB2 : wave * CBW2;
    f32x4 acc[CBW2];
    gemm_blocks<CBW2, RS2, R2T, H16 ? CO / 2 : 0>((const f32x4 *)p.w1p + lane + (long)sl * R2 * 64, Ys + l15 * YP + quad * 4, cb0, ksl * RS2, acc);
    if constexpr (H16) {                                  // unscaled partial sums from here on (exact: powers of two)
#pragma unroll
        for (int u = 0; u < CBW2; ++u) {
            acc[u] *= *(const f32x4 *)(p.w1p + (long)CN * COT + (cb0 + u) * 16 + quad * 4);
            note_not_finite(p.ovf, acc[u]);
        }
    }
    if constexpr (KS > 1) {                               // wave groups meet in LDS
        const int cn = cb0 * 16 + quad * 4;
        if (ksl > 0) *(f32x4 *)(Ps + ((ksl - 1) * BM + l15) * PP + cn) = acc[0];
        __syncthreads();
        if (ksl > 0) return;
#pragma unroll
        for (int s = 1; s < KS; ++s) acc[0] += *(const f32x4 *)(Ps + ((s - 1) * BM + l15) * PP + cn);
    }
    if constexpr (S > 1) {
        // partial products over this workgroup's k slice -> slab [sl][M][CN], written through (sc1)
        const long bytes = (long)S * p.M * CN * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.ws, 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
#pragma unroll
        for (int u = 0; u < CBW2; ++u)
            if (mok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[u]), rs,
                                                            (int)((((long)sl * p.M + m) * CN + (cb0 + u) * 16 + quad * 4) * 4), 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int *flag = (int *)Ps;
        if (tid == 0) {
            int *cnt = (int *)(p.ws + (long)S * p.M * CN) + pt;
            const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == S - 1;
            if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = last;
        }
        __syncthreads();
        if (!*flag || !mok) return;
#pragma unroll
        for (int u = 0; u < CBW2; ++u) {
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < S; ++q)
                sum += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((((long)q * p.M + m) * CN + (cb0 + u) * 16 + quad * 4) * 4), 0, 16));
            acc[u] = sum;
        }
    }
#pragma unroll
    for (int u = 0; u < CBW2; ++u) {
        const int cn = (cb0 + u) * 16 + quad * 4;
        f32x4 v = acc[u] + *(const f32x4 *)(p.b1 + cn);
        if (p.act2 == USOT_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (mok) *(f32x4 *)(p.t + (long)m * CN + cn) = v;
    }
}

template <int CN, int KSG> constexpr int ps_floats() { return KSG > 1 ? (KSG - 1) * 16 * (CN + 4) : 16; }

template <int CM, int COT, int CN, int S = 1, bool H16 = false>
__global__ __launch_bounds__(512) void pw_pair_f32_kernel(const PwF p)
{
    constexpr int NW = 8, BM = 16, XP = CM + 4, NB2 = CN / 16, KS = NB2 >= NW ? 1 : NW / NB2;
    __shared__ __attribute__((aligned(16))) float Xs[BM * XP];
    __shared__ __attribute__((aligned(16))) float Ys[BM * (COT / S + 4)];
    __shared__ __attribute__((aligned(16))) float Ps[ps_floats<CN, KS>()];
    const int tid = threadIdx.x;
    const int pt = S > 1 ? (int)blockIdx.x / S : (int)blockIdx.x, sl = S > 1 ? (int)blockIdx.x % S : 0;
    const int bm0 = pt * BM;
    // pixel tile -> LDS (rows past M are zero)
    if (p.parts > 1) {
        // deferred split-K reduction of the producing convolution: the partial tiles are summed HERE, in part order, bias and
        // ReLU applied, while the pixel tile is staged - the producer's launch ends on its k-loop (no ticket, no combine, no
        // reduction launch) and every slice of this tile does the same sum on the same values (bit-identical tiles)
        const long slab = (long)p.M * CM;
        for (int i = tid; i < BM * (CM / 4); i += NW * 64) {
            const int row = i / (CM / 4), c4 = i - row * (CM / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (bm0 + row < p.M) {
                const float *src = p.t2 + (long)(bm0 + row) * CM + c4 * 4;
                v = *(const f32x4 *)src;
                for (int q = 1; q < p.parts; ++q) v += *(const f32x4 *)(src + q * slab);
                v += *(const f32x4 *)(p.tbias + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if constexpr (H16) store_split4((char *)(Xs + row * XP), c4 * 4, CM * 2, v);
            else *(f32x4 *)(Xs + row * XP + c4 * 4) = v;
        }
    } else {
        for (int i = tid; i < BM * (CM / 4); i += NW * 64) {
            const int row = i / (CM / 4), c4 = i - row * (CM / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (bm0 + row < p.M) v = *(const f32x4 *)(p.t2 + (long)(bm0 + row) * CM + c4 * 4);
            if constexpr (H16) store_split4((char *)(Xs + row * XP), c4 * 4, CM * 2, v);
            else *(f32x4 *)(Xs + row * XP + c4 * 4) = v;
        }
    }
    __syncthreads();
    pair_tail<CM, COT, CN, S, H16>(p, Xs, Ys, Ps, pt, sl);
}

// ---- a whole bottleneck tail in one launch (layer1 at batch 1): conv2 (3x3 / stride 1, CIN -> CM) + BN + ReLU, then the
// pair above.  The 16 x 9 CIN im2col image of the pixel tile is built in LDS (zeros for padding taps), conv2's CM / 16
// column blocks run on the eight waves with the k range split over wave groups (meeting in LDS), its output tile stays in
// LDS as the pair's input.  Three launches of the frame become one.
struct PwT {
    PwF pair;                                             // t2 unused
    const float *x, *w2p, *b2;
    int H, W, OH, OW, pad_h, pad_w, dil_h, dil_w;
};

template <int CIN, int CM, int COT, int CN>
__global__ __launch_bounds__(512) void pw_triple_f32_kernel(const PwT q)
{
    constexpr int NW = 8, BM = 16, TAPS = 9, K0 = TAPS * CIN, RT0 = K0 / 16, NB0 = CM / 16;
    constexpr int KS0 = NW / NB0, RS0 = RT0 / KS0;
    static_assert(NB0 <= NW && NW % NB0 == 0 && RT0 % KS0 == 0, "conv2: one column block per wave group");
    constexpr int XP0 = K0 + 4, XP = CM + 4, NB2 = CN / 16, KS2 = NB2 >= NW ? 1 : NW / NB2;
    constexpr int PSF = ps_floats<CM, KS0>() > ps_floats<CN, KS2>() ? ps_floats<CM, KS0>() : ps_floats<CN, KS2>();
    extern __shared__ __attribute__((aligned(16))) float ldst[];
    float *X0 = ldst;                                     // [BM][XP0] im2col image
    float *Xs = X0 + BM * XP0;                            // [BM][XP]  conv2 output tile
    float *Ys = Xs + BM * XP;                             // [BM][COT + 4]
    float *Ps = Ys + BM * (COT + 4);                      // wave-group partials
    __shared__ int pix[BM][3];
    const PwF &p = q.pair;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, quad = lane >> 4;
    const int pt = blockIdx.x, bm0 = pt * BM;
    const int cb = wave % NB0, ksl = wave / NB0;
    const f32x4 *wf = (const f32x4 *)q.w2p + lane;
    GemmRing<1, RS0, RT0, (RS0 < USOT_RING ? RS0 : USOT_RING)> g;
    g.prefetch(wf, cb, ksl * RS0);
    if (tid < BM) {
        const int mm = bm0 + tid;
        if (mm < p.M) {
            const int P = q.OH * q.OW, n = mm / P, r = mm - n * P, oh = r / q.OW;
            pix[tid][0] = n * q.H;
            pix[tid][1] = oh - q.pad_h;
            pix[tid][2] = (r - oh * q.OW) - q.pad_w;
        } else {
            pix[tid][0] = -1;
        }
    }
    __syncthreads();
    for (int i = tid; i < BM * TAPS * (CIN / 4); i += NW * 64) {
        const int c4 = i % (CIN / 4), t = (i / (CIN / 4)) % TAPS, row = i / (TAPS * (CIN / 4));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pix[row][0] >= 0) {
            const int ih = pix[row][1] + (t / 3) * q.dil_h, iw = pix[row][2] + (t % 3) * q.dil_w;
            if ((unsigned)ih < (unsigned)q.H && (unsigned)iw < (unsigned)q.W)
                v = *(const f32x4 *)(q.x + ((long)(pix[row][0] + ih) * q.W + iw) * CIN + c4 * 4);
        }
        *(f32x4 *)(X0 + row * XP0 + t * CIN + c4 * 4) = v;
    }
    const f32x4 b2 = *(const f32x4 *)(q.b2 + cb * 16 + quad * 4);
    __syncthreads();
    f32x4 acc[1];
    g.run(wf, X0 + l15 * XP0 + quad * 4, cb, ksl * RS0, acc);
    if constexpr (KS0 > 1) {
        if (ksl > 0) *(f32x4 *)(Ps + ((ksl - 1) * BM + l15) * (CM + 4) + cb * 16 + quad * 4) = acc[0];
        __syncthreads();
        if (ksl == 0) {
#pragma unroll
            for (int s2 = 1; s2 < KS0; ++s2) acc[0] += *(const f32x4 *)(Ps + ((s2 - 1) * BM + l15) * (CM + 4) + cb * 16 + quad * 4);
        }
    }
    if (ksl == 0) {
        f32x4 v = acc[0] + b2;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        *(f32x4 *)(Xs + l15 * XP + cb * 16 + quad * 4) = v;           // rows past M hold relu(b2): never stored
    }
    __syncthreads();                                      // conv2's tile complete (and Ps free again)
    pair_tail<CM, COT, CN, 1>(p, Xs, Ys, Ps, pt, 0);
}

template <int CM, int CO, int CN, int S = 1, bool H16 = false> int launch(hipStream_t s, const PwF &p)
{
    hipLaunchKernelGGL((pw_pair_f32_kernel<CM, CO, CN, S, H16>), dim3(((p.M + 15) / 16) * S), dim3(512), 0, s, p);
    return hipGetLastError() == hipSuccess ? USOT_OK : USOT_ELAUNCH;
}

template <int CIN, int CM, int CO, int CN> int launch_triple(hipStream_t s, const PwT &q)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    constexpr int NB2 = CN / 16, KS2 = NB2 >= 8 ? 1 : 8 / NB2, KS0 = 8 / (CM / 16);
    constexpr int PSF = ps_floats<CM, KS0>() > ps_floats<CN, KS2>() ? ps_floats<CM, KS0>() : ps_floats<CN, KS2>();
    const size_t lds = (size_t)(16 * (9 * CIN + 4) + 16 * (CM + 4) + 16 * (CO + 4) + PSF) * sizeof(float);
    static bool raised_d[USOT_MAX_DEV] = {};
    bool &raised = raised_d[usot_dv];
    if (lds > 64 * 1024 && !raised) {
        if (hipFuncSetAttribute((const void *)pw_triple_f32_kernel<CIN, CM, CO, CN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return USOT_ELAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((pw_triple_f32_kernel<CIN, CM, CO, CN>), dim3((q.pair.M + 15) / 16), dim3(512), lds, s, q);
    return hipGetLastError() == hipSuccess ? USOT_OK : USOT_ELAUNCH;
}

// channel slices per pixel tile (1 = unsliced), by shape and number of pixel tiles: layer2 / layer3 at batch 1 have 61
// pixel tiles for 256 CUs
int slices(int M, int CM, int CN)
{
    if (CM == 128 && CN == 128 && M <= 2 * 961) return 4;
    if (CM == 256 && CN == 256 && M <= 1200) return 4;
    return 1;
}

// ---- one pointwise convolution in the same style, for layer3's 1x1 layers at batch 1 (M = 961: 61 pixel tiles): the
// 16 x K pixel tile in LDS, N / 16 / S column blocks per workgroup (S workgroups per pixel tile, no exchange: each owns
// its output channels), filters streamed in fragment order.  When a workgroup has fewer column blocks than waves the k
// range is split over wave groups meeting in LDS.  Y = act(X . W^T + b (+ R)).
struct Pw1 {
    const float *x, *wp, *b, *res;
    float *y;
    int M, act;
};

template <int K, int N, int S, int PFD = USOT_RING>
__global__ __launch_bounds__(512) void pw_single_f32_kernel(const Pw1 p)
{
    constexpr int NW = 8, BM = 16;
    constexpr int RT = K / 16, NB = N / 16 / S;           // rounds, column blocks of this workgroup
    constexpr int KS = NB >= NW ? 1 : NW / NB, CBW = NB >= NW ? NB / NW : 1, RS = RT / KS;
    static_assert((NB % NW == 0 || NW % NB == 0) && RT % KS == 0, "shape");
    constexpr int XP = K + 4, PP = NB * 16 + 4;
    extern __shared__ __attribute__((aligned(16))) float lds1[];
    float *Xs = lds1;                                     // [BM][XP]
    float *Ps = lds1 + BM * XP;                           // [(KS - 1)][BM][PP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, quad = lane >> 4;
    const int pt = (int)blockIdx.x / S, sl = (int)blockIdx.x % S;
    const int bm0 = pt * BM, m = bm0 + l15;
    const bool mok = m < p.M;
    const int ksl = KS > 1 ? wave / NB : 0;
    const int cb0 = KS > 1 ? wave % NB : wave * CBW;      // within the slice
    const f32x4 *wf = (const f32x4 *)p.wp + lane + (long)sl * NB * RT * 64;
    GemmRing<CBW, RS, RT, (CBW * RS < PFD ? CBW * RS : PFD)> g;
    g.prefetch(wf, cb0, ksl * RS);                        // the filter stream starts before the pixel tile arrives
    for (int i = tid; i < BM * (K / 4); i += NW * 64) {
        const int row = i / (K / 4), c4 = i - row * (K / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (bm0 + row < p.M) v = *(const f32x4 *)(p.x + (long)(bm0 + row) * K + c4 * 4);
        *(f32x4 *)(Xs + row * XP + c4 * 4) = v;
    }
    f32x4 rr[CBW], bb[CBW];
#pragma unroll
    for (int u = 0; u < CBW; ++u) {
        const int n = (sl * NB + cb0 + u) * 16 + quad * 4;
        rr[u] = (mok && p.res && ksl == 0) ? *(const f32x4 *)(p.res + (long)m * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        bb[u] = *(const f32x4 *)(p.b + n);
    }
    __syncthreads();
    f32x4 acc[CBW];
    g.run(wf, Xs + l15 * XP + quad * 4, cb0, ksl * RS, acc);
    if constexpr (KS > 1) {
        if (ksl > 0) *(f32x4 *)(Ps + ((ksl - 1) * BM + l15) * PP + cb0 * 16 + quad * 4) = acc[0];
        __syncthreads();
        if (ksl > 0) return;
#pragma unroll
        for (int s2 = 1; s2 < KS; ++s2) acc[0] += *(const f32x4 *)(Ps + ((s2 - 1) * BM + l15) * PP + cb0 * 16 + quad * 4);
    }
#pragma unroll
    for (int u = 0; u < CBW; ++u) {
        f32x4 v = acc[u] + bb[u] + rr[u];
        if (p.act == USOT_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (mok) *(f32x4 *)(p.y + (long)m * N + (sl * NB + cb0 + u) * 16 + quad * 4) = v;
    }
}

template <int K, int N, int S> int launch1(hipStream_t s, const Pw1 &p)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    constexpr int NB = N / 16 / S, KS = NB >= 8 ? 1 : 8 / NB;
    const size_t lds = (size_t)(16 * (K + 4) + (KS > 1 ? (KS - 1) * 16 * (NB * 16 + 4) : 0)) * sizeof(float);
    if (lds > 64 * 1024) {
        static bool raised_d[USOT_MAX_DEV] = {};
    bool &raised = raised_d[usot_dv];
        if (!raised) {
            if (hipFuncSetAttribute((const void *)pw_single_f32_kernel<K, N, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return USOT_ELAUNCH;
            raised = true;
        }
    }
    hipLaunchKernelGGL((pw_single_f32_kernel<K, N, S>), dim3(((p.M + 15) / 16) * S), dim3(512), lds, s, p);
    return hipGetLastError() == hipSuccess ? USOT_OK : USOT_ELAUNCH;
}

// ---- the same streaming form for 3x3 convolutions at small M (layer3 / layer2's conv2 at batch 1): the 16-pixel tile is
// the im2col image of the pixels (16 x 9 Cin floats: 147 KB for Cin = 256, one workgroup per CU), built once from the
// NHWC map with zeros for padding taps; k = (kh, kw, ci) as in the packed filter bank.  Stride 1.
struct PwC {
    const float *x, *wp, *b, *res;
    float *y;
    int M, H, W, OH, OW, pad_h, pad_w, dil_h, dil_w, act;
};

template <int CIN, int N, int S>
__global__ __launch_bounds__(512) void stream_conv3x3_f32_kernel(const PwC p)
{
    constexpr int NW = 8, BM = 16, TAPS = 9, K = TAPS * CIN;
    constexpr int RT = K / 16, NB = N / 16 / S;
    constexpr int KS = NB >= NW ? 1 : NW / NB, CBW = NB >= NW ? NB / NW : 1, RS = RT / KS;
    static_assert((NB % NW == 0 || NW % NB == 0) && RT % KS == 0, "shape");
    constexpr int XP = K + 4, PP = NB * 16 + 4;
    extern __shared__ __attribute__((aligned(16))) float ldsc[];
    float *Xs = ldsc;                                     // [BM][XP]
    float *Ps = ldsc + BM * XP;                           // [(KS - 1)][BM][PP]
    __shared__ int pix[BM][3];                            // (row base of the image, oh - pad, ow - pad) per pixel, -1 = past M

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, quad = lane >> 4;
    const int pt = (int)blockIdx.x / S, sl = (int)blockIdx.x % S;
    const int bm0 = pt * BM, m = bm0 + l15;
    const bool mok = m < p.M;
    const int ksl = KS > 1 ? wave / NB : 0;
    const int cb0 = KS > 1 ? wave % NB : wave * CBW;
    const f32x4 *wf = (const f32x4 *)p.wp + lane + (long)sl * NB * RT * 64;
    GemmRing<CBW, RS, RT, (CBW * RS < USOT_RING ? CBW * RS : USOT_RING)> g;
    g.prefetch(wf, cb0, ksl * RS);
    if (tid < BM) {
        const int mm = bm0 + tid;
        if (mm < p.M) {
            const int P = p.OH * p.OW, n = mm / P, r = mm - n * P, oh = r / p.OW;
            pix[tid][0] = n * p.H;
            pix[tid][1] = oh - p.pad_h;
            pix[tid][2] = (r - oh * p.OW) - p.pad_w;
        } else {
            pix[tid][0] = -1;
        }
    }
    __syncthreads();
    for (int i = tid; i < BM * TAPS * (CIN / 4); i += NW * 64) {
        const int c4 = i % (CIN / 4), t = (i / (CIN / 4)) % TAPS, row = i / (TAPS * (CIN / 4));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pix[row][0] >= 0) {
            const int ih = pix[row][1] + (t / 3) * p.dil_h, iw = pix[row][2] + (t % 3) * p.dil_w;
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                v = *(const f32x4 *)(p.x + ((long)(pix[row][0] + ih) * p.W + iw) * CIN + c4 * 4);
        }
        *(f32x4 *)(Xs + row * XP + t * CIN + c4 * 4) = v;
    }
    f32x4 rr[CBW], bb[CBW];
#pragma unroll
    for (int u = 0; u < CBW; ++u) {
        const int n = (sl * NB + cb0 + u) * 16 + quad * 4;
        rr[u] = (mok && p.res && ksl == 0) ? *(const f32x4 *)(p.res + (long)m * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        bb[u] = *(const f32x4 *)(p.b + n);
    }
    __syncthreads();
    f32x4 acc[CBW];
    g.run(wf, Xs + l15 * XP + quad * 4, cb0, ksl * RS, acc);
    if constexpr (KS > 1) {
        if (ksl > 0) *(f32x4 *)(Ps + ((ksl - 1) * BM + l15) * PP + cb0 * 16 + quad * 4) = acc[0];
        __syncthreads();
        if (ksl > 0) return;
#pragma unroll
        for (int s2 = 1; s2 < KS; ++s2) acc[0] += *(const f32x4 *)(Ps + ((s2 - 1) * BM + l15) * PP + cb0 * 16 + quad * 4);
    }
#pragma unroll
    for (int u = 0; u < CBW; ++u) {
        f32x4 v = acc[u] + bb[u] + rr[u];
        if (p.act == USOT_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (mok) *(f32x4 *)(p.y + (long)m * N + (sl * NB + cb0 + u) * 16 + quad * 4) = v;
    }
}

template <int CIN, int N, int S> int launch3(hipStream_t s, const PwC &p)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    constexpr int K = 9 * CIN, NB = N / 16 / S, KS = NB >= 8 ? 1 : 8 / NB;
    const size_t lds = (size_t)(16 * (K + 4) + (KS > 1 ? (KS - 1) * 16 * (NB * 16 + 4) : 0)) * sizeof(float);
    static bool raised_d[USOT_MAX_DEV] = {};
    bool &raised = raised_d[usot_dv];
    if (lds > 64 * 1024 && !raised) {
        if (hipFuncSetAttribute((const void *)stream_conv3x3_f32_kernel<CIN, N, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return USOT_ELAUNCH;
        raised = true;
    }
    hipLaunchKernelGGL((stream_conv3x3_f32_kernel<CIN, N, S>), dim3(((p.M + 15) / 16) * S), dim3(512), lds, s, p);
    return hipGetLastError() == hipSuccess ? USOT_OK : USOT_ELAUNCH;
}

}  // namespace

extern "C" int usot_pw_pair_f32_supported(int CM, int CO, int CN)
{
    return (CM == 64 && CO == 256 && (CN == 64 || CN == 128)) || (CM == 128 && CO == 512 && (CN == 128 || CN == 256)) ||
           (CM == 256 && CO == 1024 && CN == 256);
}

/* workspace of the sliced form in floats: S slabs [M][CN] + one ticket word per pixel tile; 0 = none needed.  It must be
 * zero before the first launch (the tickets are reset by the last arriver of every launch). */
extern "C" int64_t usot_pw_pair_f32_ws_floats(int M, int CM, int CO, int CN)
{
    (void)CO;
    const int S = slices(M, CM, CN);
    return S > 1 ? (int64_t)S * M * CN + (M + 15) / 16 : 0;
}

static int pw_pair_f32_launch(void *stream, const usot_pw_pair_desc *d, bool split16);

extern "C" int usot_pw_pair_f32(void *stream, const usot_pw_pair_desc *d) { return pw_pair_f32_launch(stream, d, false); }

/* The same pair on SPLIT-fp16 operands (shapes: usot_pw_pair_f32s_supported): d->w3p / d->w1 are the banks pre-split by
 * usot_amd/hip.py: pw_pair_s16_pack - per column block of 16 rows and 32-k step a hi fragment then a lo fragment, followed by the
 * bank's per-row factors 1 / (row scale x 8); everything else as usot_pw_pair_f32.  Values staged into LDS must be below 8 188. */
extern "C" int usot_pw_pair_f32s(void *stream, const usot_pw_pair_desc *d) { return pw_pair_f32_launch(stream, d, true); }

extern "C" int usot_pw_pair_f32s_supported(int CM, int CO, int CN) { return CM == 256 && CO == 1024 && CN == 256; }

static int pw_pair_f32_launch(void *stream, const usot_pw_pair_desc *d, bool split16)
{
    if (!d || !d->t2 || !d->w3p || !d->b3 || !d->res || !d->w1 || !d->b1 || !d->y || !d->t || d->M <= 0) return USOT_EINVAL;
    if (d->act2 != USOT_ACT_NONE && d->act2 != USOT_ACT_RELU) return USOT_EINVAL;
    const uintptr_t al = (uintptr_t)d->t2 | (uintptr_t)d->w3p | (uintptr_t)d->b3 | (uintptr_t)d->res | (uintptr_t)d->w1 |
                         (uintptr_t)d->b1 | (uintptr_t)d->y | (uintptr_t)d->t | (uintptr_t)d->ws;
    if (al & 15) return USOT_EINVAL;
    if (d->t2_parts > 1 && (!d->t2_bias || ((uintptr_t)d->t2_bias & 15) || d->t2_parts > 16)) return USOT_EINVAL;
    PwF p{(const float *)d->t2, (const float *)d->w3p, d->b3, (const float *)d->res, (const float *)d->w1, d->b1,
          (float *)d->y, (float *)d->t, (float *)d->ws, d->M, d->act2};
    p.parts = d->t2_parts > 1 ? d->t2_parts : 0;
    p.tbias = d->t2_bias;
    if (d->res_parts > 1 && (!d->res_bias || ((uintptr_t)d->res_bias & 15) || d->res_parts > 16)) return USOT_EINVAL;
    p.rparts = d->res_parts > 1 ? d->res_parts : 0;
    p.rbias = d->res_bias;
    if ((uintptr_t)d->ovf & 3) return USOT_EINVAL;
    p.ovf = split16 ? d->ovf : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const bool sliced = slices(d->M, d->CM, d->CN) > 1 && d->ws;          /* no workspace: the unsliced form */
    if (split16) {
        if (d->CM == 256 && d->CO == 1024 && d->CN == 256 && sliced) return launch<256, 1024, 256, 4, true>(s, p);
        return USOT_EINVAL;
    }
    if (d->CM == 64 && d->CO == 256 && d->CN == 64) return launch<64, 256, 64>(s, p);
    if (d->CM == 64 && d->CO == 256 && d->CN == 128) return launch<64, 256, 128>(s, p);
    if (d->CM == 128 && d->CO == 512 && d->CN == 128) return sliced ? launch<128, 512, 128, 4>(s, p) : launch<128, 512, 128>(s, p);
    if (d->CM == 128 && d->CO == 512 && d->CN == 256) return launch<128, 512, 256>(s, p);
    if (d->CM == 256 && d->CO == 1024 && d->CN == 256 && sliced) return launch<256, 1024, 256, 4>(s, p);   /* sliced only: an unsliced Y tile (64 KB) does not fit */
    return USOT_EINVAL;
}

/* one pointwise convolution, small M: y[M][N] = act(x[M][K] . W^T + b (+ res)); wp = W in the fragment order above.
 * Shapes (K, N): (1024, 256), (256, 1024), (512, 128), (128, 512). */
extern "C" int usot_pw_single_f32_supported(int K, int N)
{
    return (K == 1024 && N == 256) || (K == 256 && N == 1024) || (K == 512 && N == 128) || (K == 128 && N == 512);
}

extern "C" int usot_pw_single_f32(void *stream, const float *x, const float *wp, const float *b, const float *res, float *y,
                                  int M, int K, int N, int act)
{
    if (!x || !wp || !b || !y || M <= 0 || (act != USOT_ACT_NONE && act != USOT_ACT_RELU)) return USOT_EINVAL;
    if (((uintptr_t)x | (uintptr_t)wp | (uintptr_t)b | (uintptr_t)res | (uintptr_t)y) & 15) return USOT_EINVAL;
    const Pw1 p{x, wp, b, res, y, M, act};
    hipStream_t s = (hipStream_t)stream;
    if (K == 1024 && N == 256) return launch1<1024, 256, 4>(s, p);
    if (K == 256 && N == 1024) return launch1<256, 1024, 8>(s, p);
    if (K == 512 && N == 128) return launch1<512, 128, 2>(s, p);
    if (K == 128 && N == 512) return launch1<128, 512, 4>(s, p);
    return USOT_EINVAL;
}

/* 3x3 / stride 1 convolution, small M, same streaming form: x NHWC [Nb][H][W][Cin] dense, wp = the packed bank
 * [N][9 * Cin] ((kh, kw, ci) order) in fragment order, y [M = Nb * OH * OW][N].  Shapes (Cin, N): (256, 256), (128, 128). */
extern "C" int usot_stream_conv3x3_f32_supported(int Cin, int N) { return (Cin == 256 && N == 256) || (Cin == 128 && N == 128); }

extern "C" int usot_stream_conv3x3_f32(void *stream, const float *x, const float *wp, const float *b, const float *res, float *y,
                                       int Nb, int H, int W, int Cin, int OH, int OW, int N, int pad_h, int pad_w, int dil_h, int dil_w,
                                       int act)
{
    if (!x || !wp || !b || !y || Nb <= 0 || (act != USOT_ACT_NONE && act != USOT_ACT_RELU)) return USOT_EINVAL;
    if (((uintptr_t)x | (uintptr_t)wp | (uintptr_t)b | (uintptr_t)res | (uintptr_t)y) & 15) return USOT_EINVAL;
    if (OH != H + 2 * pad_h - 2 * dil_h || OW != W + 2 * pad_w - 2 * dil_w || OH <= 0 || OW <= 0) return USOT_EINVAL;
    const PwC p{x, wp, b, res, y, Nb * OH * OW, H, W, OH, OW, pad_h, pad_w, dil_h, dil_w, act};
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 256 && N == 256) return launch3<256, 256, 4>(s, p);
    if (Cin == 128 && N == 128) return launch3<128, 128, 4>(s, p);
    return USOT_EINVAL;
}

/* conv2 (3x3 / stride 1) + BN + ReLU, conv3 + BN + residual + ReLU and the next block's conv1 + BN + act2 in ONE launch
 * (layer1 of the fp32 frame): x NHWC [Nb][H][W][Cin]; w2p the packed conv2 bank [CM][9 Cin] in fragment order; the rest as
 * usot_pw_pair_f32 (d->t2 is ignored, d->M = Nb * OH * OW).  Shapes (Cin, CM, CO, CN): (64, 64, 256, 64 | 128), (128, 128, 512, 128). */
extern "C" int usot_pw_triple_f32_supported(int Cin, int CM, int CO, int CN)
{
    return (Cin == 64 && CM == 64 && CO == 256 && (CN == 64 || CN == 128)) || (Cin == 128 && CM == 128 && CO == 512 && CN == 128);
}

extern "C" int usot_pw_triple_f32(void *stream, const float *x, const float *w2p, const float *b2, const usot_pw_pair_desc *d,
                                  int Nb, int H, int W, int Cin, int OH, int OW, int pad_h, int pad_w, int dil_h, int dil_w)
{
    if (!x || !w2p || !b2 || !d || !d->w3p || !d->b3 || !d->res || !d->w1 || !d->b1 || !d->y || !d->t) return USOT_EINVAL;
    if (!usot_pw_triple_f32_supported(Cin, d->CM, d->CO, d->CN) || d->M != Nb * OH * OW || d->M <= 0) return USOT_EINVAL;
    if (OH != H + 2 * pad_h - 2 * dil_h || OW != W + 2 * pad_w - 2 * dil_w) return USOT_EINVAL;
    if (d->act2 != USOT_ACT_NONE && d->act2 != USOT_ACT_RELU) return USOT_EINVAL;
    const uintptr_t al = (uintptr_t)x | (uintptr_t)w2p | (uintptr_t)b2 | (uintptr_t)d->w3p | (uintptr_t)d->b3 | (uintptr_t)d->res |
                         (uintptr_t)d->w1 | (uintptr_t)d->b1 | (uintptr_t)d->y | (uintptr_t)d->t;
    if (al & 15) return USOT_EINVAL;
    PwT q;
    q.pair = PwF{nullptr, (const float *)d->w3p, d->b3, (const float *)d->res, (const float *)d->w1, d->b1, (float *)d->y,
                 (float *)d->t, nullptr, d->M, d->act2};
    q.x = x; q.w2p = w2p; q.b2 = b2;
    q.H = H; q.W = W; q.OH = OH; q.OW = OW; q.pad_h = pad_h; q.pad_w = pad_w; q.dil_h = dil_h; q.dil_w = dil_w;
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 128) return launch_triple<128, 128, 512, 128>(s, q);
    if (d->CN == 64) return launch_triple<64, 64, 256, 64>(s, q);
    return launch_triple<64, 64, 256, 128>(s, q);
}
