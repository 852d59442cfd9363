// Fused pair of pointwise convolutions for the fp32 batch-1 frame: a bottleneck's conv3 + BN + residual + ReLU
// (modules.py:48-56) and the NEXT block's conv1 + BN + ReLU (modules.py:40-42) in one launch.
//
// Why: at batch 1 layer1 / layer2's 1x1 convolutions are 0.1-0.25 GFLOP each and cost 9-10.5 us per launch — launch,
// one cold round trip for the activations the previous kernel wrote, the boundary — for ~1 us of matrix work
// (profiles/: layer1 + layer2 = 29 % of the frame at 16-29 TFLOP/s).  The pair shares its pixel tile: Y never leaves
// the CU between the two GEMMs, one launch and one cold round trip disappear per pair.
//
//   Y[M][CO] = relu(T2[M][CM] . W3^T + b3 + R[M][CO])        (stored: it is the next block's residual)
//   T [M][CN] = act2(Y . W1^T + b1)
//
// One workgroup = 16 pixels (M = 3969 -> 249 workgroups, M = 961 -> 61), eight wavefronts.  v_mfma_f32_16x16x4_f32
// with the conv kernel's operand roles (filters = A, pixels = B: a lane's accumulator is 4 consecutive channels of one
// pixel, 16-byte epilogue accesses) and its ds_read_b128 trick (a lane reads 4 consecutive k and issues 4 MFMAs; quad q
// owns k-slot q).  Both filter banks are pre-packed in FRAGMENT order (one contiguous KiB per 16 channels x 16 k, see
// usot_pw_pair_f32 in usot_hip.h) and stream straight from L2 into registers, two column blocks / eight rounds ahead;
// only the pixel tile and Y go through LDS.  GEMM2 has CN / 16 column blocks for eight waves: when that is 4 the k
// range is split in two and the halves meet in LDS (fixed order).
// S > 1 (layer2 at batch 1: only 61 pixel tiles): S workgroups share a pixel tile, each owning CO / S channels of Y —
// GEMM1 for its slice, then the partial product of GEMM2 over that slice of k — and meet through a workspace exactly as
// the conv kernels' in-launch split-K does: write-through slab stores, one relaxed ticket per pixel tile, the last
// arriver sums the S slabs in slice order and applies bias / activation (no fences; tickets are zero before the first
// launch and reset by the last arriver).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PwF {
    const float *t2, *w3p, *b3, *res, *w1p, *b1;
    float *y, *t, *ws;
    int M, act2;
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int CM, int COT, int CN, int S = 1>
__global__ __launch_bounds__(512) void pw_pair_f32_kernel(const PwF p)
{
    constexpr int NW = 8, BM = 16;
    constexpr int CO = COT / S;                           // channels of Y this workgroup owns
    constexpr int R1 = CM / 16, NB1 = CO / 16;            // GEMM1: rounds of 16 k, column blocks of 16 channels
    constexpr int R2 = CO / 16, NB2 = CN / 16;
    constexpr int KS = NB2 >= NW ? 1 : NW / NB2;          // k-slices of GEMM2
    constexpr int CBW2 = NB2 >= NW ? NB2 / NW : 1;        // column blocks per wave in GEMM2
    static_assert(NB1 % NW == 0 && (NB2 % NW == 0 || NW % NB2 == 0) && R2 % KS == 0, "shape");
    constexpr int XP = CM + 4, YP = CO + 4, PP = CN + 4;  // LDS row pitches (floats): 16 B aligned, rows spread over banks
    __shared__ __attribute__((aligned(16))) float Xs[BM * XP];
    __shared__ __attribute__((aligned(16))) float Ys[BM * YP];
    __shared__ __attribute__((aligned(16))) float Ps[(KS > 1 ? (KS - 1) : 1) * BM * PP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, quad = lane >> 4;
    const int pt = S > 1 ? (int)blockIdx.x / S : (int)blockIdx.x, sl = S > 1 ? (int)blockIdx.x % S : 0;
    const int bm0 = pt * BM;
    const int m = bm0 + l15;
    const bool mok = m < p.M;

    // filter fragments of the first column block(s) start flying before anything else
    const f32x4 *w3 = (const f32x4 *)p.w3p + lane + (long)sl * NB1 * R1 * 64;      // fragment (cb, r) at [(cb * R1 + r) * 64]
    f32x4 wa[2][R1];
#pragma unroll
    for (int r = 0; r < R1; ++r) wa[0][r] = w3[(wave * R1 + r) * 64];

    // pixel tile -> LDS (rows past M are zero)
    for (int i = tid; i < BM * (CM / 4); i += NW * 64) {
        const int row = i / (CM / 4), c4 = i - row * (CM / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (bm0 + row < p.M) v = *(const f32x4 *)(p.t2 + (long)(bm0 + row) * CM + c4 * 4);
        *(f32x4 *)(Xs + row * XP + c4 * 4) = v;
    }
    __syncthreads();
    f32x4 xb[R1];
#pragma unroll
    for (int r = 0; r < R1; ++r) xb[r] = *(const f32x4 *)(Xs + l15 * XP + r * 16 + quad * 4);

    // ---- GEMM1: column blocks wave, wave + 8, ...
#pragma unroll
    for (int i = 0; i < NB1 / NW; ++i) {
        const int cb = wave + i * NW;
        const int co = cb * 16 + quad * 4;                 // within the slice
        const int cog = sl * CO + co;
        // residual and bias of this block, and the next block's fragments, fly under the MFMAs
        f32x4 rr = {0.f, 0.f, 0.f, 0.f};
        if (mok) rr = *(const f32x4 *)(p.res + (long)m * COT + cog);
        const f32x4 bb = *(const f32x4 *)(p.b3 + cog);
        if (i + 1 < NB1 / NW) {
#pragma unroll
            for (int r = 0; r < R1; ++r) wa[(i + 1) & 1][r] = w3[((cb + NW) * R1 + r) * 64];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R1; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i & 1][r][c], xb[r][c], acc, 0, 0, 0);
        f32x4 v = acc + bb + rr;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        if (mok) *(f32x4 *)(p.y + (long)m * COT + cog) = v;
        *(f32x4 *)(Ys + l15 * YP + co) = v;
    }

    // ---- GEMM2: K = CO from LDS; wave -> (column block(s), k-slice)
    const int ksl = KS > 1 ? wave / NB2 : 0;
    const int cb0 = KS > 1 ? wave % NB2 : wave * CBW2;
    constexpr int RS = R2 / KS;                           // rounds per slice
    constexpr int PF = RS < 8 ? RS : 8;                   // filter fragments in flight
    constexpr int R2T = COT / 16;                         // rounds of a whole filter row
    const f32x4 *w1 = (const f32x4 *)p.w1p + lane + (long)sl * R2 * 64;
    f32x4 wb[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) wb[j] = w1[(cb0 * R2T + ksl * RS + j) * 64];
    __syncthreads();                                      // Y tile complete
#pragma unroll
    for (int u = 0; u < CBW2; ++u) {
        const int cb = cb0 + u;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r0 = 0; r0 < RS; r0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int r = ksl * RS + r0 + j;
                const f32x4 yb = *(const f32x4 *)(Ys + l15 * YP + r * 16 + quad * 4);
                const f32x4 a = wb[j];
                // refill this slot with the fragment PF rounds ahead (of this column block, then of the next)
                const int nr = r0 + j + PF;
                if (nr < RS) wb[j] = w1[(cb * R2T + ksl * RS + nr) * 64];
                else if (u + 1 < CBW2) wb[j] = w1[((cb + 1) * R2T + ksl * RS + (nr - RS)) * 64];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], yb[c], acc, 0, 0, 0);
            }
        }
        const int cn = cb * 16 + quad * 4;
        if (KS > 1) {
            if (ksl > 0) *(f32x4 *)(Ps + ((ksl - 1) * BM + l15) * PP + cn) = acc;
            __syncthreads();
            if (ksl > 0) continue;
#pragma unroll
            for (int s = 1; s < KS; ++s) acc += *(const f32x4 *)(Ps + ((s - 1) * BM + l15) * PP + cn);
        }
        if constexpr (S > 1) {
            // partial product over this workgroup's k slice -> slab [sl][M][CN], written through (sc1)
            static_assert(KS == 1 && CBW2 == 1, "one column block per wave in the sliced form");
            const long bytes = (long)S * p.M * CN * 4;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.ws, 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
            if (mok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc), rs, (int)((((long)sl * p.M + m) * CN + cn) * 4), 0, 16);
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
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < S; ++q)
                sum += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((((long)q * p.M + m) * CN + cn) * 4), 0, 16));
            acc = sum;
        }
        f32x4 v = acc + *(const f32x4 *)(p.b1 + cn);
        if (p.act2 == USOT_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (mok) *(f32x4 *)(p.t + (long)m * CN + cn) = v;
    }
}

template <int CM, int CO, int CN, int S = 1> int launch(hipStream_t s, const PwF &p)
{
    hipLaunchKernelGGL((pw_pair_f32_kernel<CM, CO, CN, S>), dim3(((p.M + 15) / 16) * S), dim3(512), 0, s, p);
    return hipGetLastError() == hipSuccess ? USOT_OK : USOT_ELAUNCH;
}

// channel slices per pixel tile: layer2's pairs at batch 1-2 have 61-121 pixel tiles for 256 CUs
int slices(int M, int CM, int CN) { return (CM == 128 && CN == 128 && M <= 2 * 961) ? 4 : 1; }

}  // namespace

extern "C" int usot_pw_pair_f32_supported(int CM, int CO, int CN)
{
    return (CM == 64 && CO == 256 && (CN == 64 || CN == 128)) || (CM == 128 && CO == 512 && (CN == 128 || CN == 256));
}

/* workspace of the sliced form in floats: S slabs [M][CN] + one ticket word per pixel tile; 0 = none needed.  It must be
 * zero before the first launch (the tickets are reset by the last arriver of every launch). */
extern "C" int64_t usot_pw_pair_f32_ws_floats(int M, int CM, int CO, int CN)
{
    (void)CO;
    const int S = slices(M, CM, CN);
    return S > 1 ? (int64_t)S * M * CN + (M + 15) / 16 : 0;
}

extern "C" int usot_pw_pair_f32(void *stream, const usot_pw_pair_desc *d)
{
    if (!d || !d->t2 || !d->w3p || !d->b3 || !d->res || !d->w1 || !d->b1 || !d->y || !d->t || d->M <= 0) return USOT_EINVAL;
    if (d->act2 != USOT_ACT_NONE && d->act2 != USOT_ACT_RELU) return USOT_EINVAL;
    const uintptr_t al = (uintptr_t)d->t2 | (uintptr_t)d->w3p | (uintptr_t)d->b3 | (uintptr_t)d->res | (uintptr_t)d->w1 |
                         (uintptr_t)d->b1 | (uintptr_t)d->y | (uintptr_t)d->t;
    if (al & 15) return USOT_EINVAL;
    const PwF p{(const float *)d->t2, (const float *)d->w3p, d->b3, (const float *)d->res, (const float *)d->w1, d->b1,
                (float *)d->y, (float *)d->t, (float *)d->ws, d->M, d->act2};
    hipStream_t s = (hipStream_t)stream;
    const bool sliced = slices(d->M, d->CM, d->CN) > 1 && d->ws;          /* no workspace: the unsliced form */
    if (sliced && d->CM == 128 && d->CO == 512 && d->CN == 128) return launch<128, 512, 128, 4>(s, p);
    if (d->CM == 64 && d->CO == 256 && d->CN == 64) return launch<64, 256, 64>(s, p);
    if (d->CM == 64 && d->CO == 256 && d->CN == 128) return launch<64, 256, 128>(s, p);
    if (d->CM == 128 && d->CO == 512 && d->CN == 128) return launch<128, 512, 128>(s, p);
    if (d->CM == 128 && d->CO == 512 && d->CN == 256) return launch<128, 512, 256>(s, p);
    return USOT_EINVAL;
}
