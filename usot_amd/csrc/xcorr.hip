// Depthwise cross-correlation kernels (reference lib/models/connect.py:86-102,147-157).
//
// Depthwise xcorr does 25 MACs per output against ~5 bytes of compulsory traffic: it is
// HBM/L2-bound, not a dense contraction, so neither kernel touches MFMA.
//
// (1) groupdw_nhwc: the engine's hot path.  Fuses the three xcorrs of GroupDW
//     (5x5, 3x5, 5x3 templates) and the softmax(weight)-weighted sum into one pass over
//     NHWC maps: lanes are channels (256-byte coalesced rows), every thread slides a
//     5-row accumulator window down its column strip, so each search element is fetched
//     once per strip and no intermediate map is ever written.  Template taps are scaled
//     by the branch weight once, in registers.
// (2) xcorr_planes: drop-in for `xcorr_depthwise` on NCHW planes (API edge).  One
//     wavefront per plane (two when Wx <= 32): lane = column; the template tile sits in
//     LDS and is read by broadcast; the horizontal window taps come from neighbouring
//     lanes by wavefront shuffles, computed once per input row and kept in a rolling
//     register window.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "usot_hip.h"
#include "common.h"

namespace {

// -------------------------------------------------------------------------------------
// (1) fused GroupDW, NHWC
// -------------------------------------------------------------------------------------
// One launch covers up to three "segments" (the cls, reg and memory GroupDWs of a frame:
// different template sets / channel halves / branch weights, same geometry), so a frame
// pays one launch instead of three.  A thread owns one channel (lane = channel: every
// global access of a wavefront is a contiguous 256-byte row) and an RS x CW patch of
// output pixels; the 55 template taps live in registers pre-scaled by softmax(weight), each
// search row of the patch is loaded once and feeds up to 5 output rows x 5 taps
// (5.2 FMA per load).  Loop bounds are compile-time: no accumulator rotation, the
// compiler is free to hoist the next row's loads above the current row's FMAs.
struct GdwSeg {
    const float *x[3];
    const float *z[3];
    float *out;
    int x_cs[3], x_co[3], z_cs[3], z_co[3];
    float wsm[3];
    int S, x_rep;
};
struct GdwK {
    GdwSeg seg[3];
    int nseg, OH, OW, C;
    int nty, ntx;      // patch grid
    int total;         // samples over all segments
};

template <int RS, int CW, int WPS>
__global__ __launch_bounds__(256, WPS) void groupdw_nhwc_kernel(const GdwK p)
{
    // Work unit = (sample, patch); channel group = 64 lanes.  With C = 256 the 1-D grid is
    // laid out so that XCD x (= block id % 8, observed dispatch order; speed only) works on
    // channel group x % 4 of the samples with parity x / 4: all 25 patches of one
    // (sample, channel group) meet in ONE XCD's L2, so the 3.2x halo re-reads between
    // neighbouring patches are L2 hits and HBM sees each search element once.
    const int ntile = p.nty * p.ntx;
    const int wave = threadIdx.x >> 6;
    int cgi, s, tile;
    if (p.C == 256) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int half = xcd >> 2;
        cgi = xcd & 3;
        const int v = slot * 4 + wave;
        const int sl = v / ntile;
        tile = v - sl * ntile;
        s = sl * 2 + half;
    } else {
        const int ncg = p.C >> 6;
        cgi = blockIdx.x % ncg;
        const int v = (blockIdx.x / ncg) * 4 + wave;
        s = v / ntile;
        tile = v - s * ntile;
    }
    if (s >= p.total) return;
    const int c = cgi * 64 + (threadIdx.x & 63);
    int sg = 0;
    while (sg + 1 < p.nseg && s >= p.seg[sg].S) { s -= p.seg[sg].S; ++sg; }
    const GdwSeg &g = p.seg[sg];
    const int i0 = (tile / p.ntx) * RS, j0 = (tile % p.ntx) * CW;
    const int xs = s / g.x_rep;

    float k0[5][5], k1[3][5], k2[5][3];
    {
        const float *z0 = g.z[0] + (long)s * 25 * g.z_cs[0] + g.z_co[0] + c;
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int v = 0; v < 5; ++v) k0[u][v] = g.wsm[0] * z0[(u * 5 + v) * g.z_cs[0]];
        const float *z1 = g.z[1] + (long)s * 15 * g.z_cs[1] + g.z_co[1] + c;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 5; ++v) k1[u][v] = g.wsm[1] * z1[(u * 5 + v) * g.z_cs[1]];
        const float *z2 = g.z[2] + (long)s * 15 * g.z_cs[2] + g.z_co[2] + c;
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) k2[u][v] = g.wsm[2] * z2[(u * 3 + v) * g.z_cs[2]];
    }
    float acc[RS][CW];
#pragma unroll
    for (int i = 0; i < RS; ++i)
#pragma unroll
        for (int j = 0; j < CW; ++j) acc[i][j] = 0.f;

    // rows/cols past the map edge (partial patches only) are clamped: they feed only outputs
    // that are never stored
    {   // 5x5 branch
        const int H = p.OH + 4, W = p.OW + 4;
        const float *x = g.x[0] + (long)xs * H * W * g.x_cs[0] + g.x_co[0] + c;
#pragma unroll
        for (int rr = 0; rr < RS + 4; ++rr) {
            const int row = min(i0 + rr, H - 1);
            float xv[CW + 4];
#pragma unroll
            for (int q = 0; q < CW + 4; ++q) xv[q] = x[((long)row * W + min(j0 + q, W - 1)) * g.x_cs[0]];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int i = rr - u;
                if (i < 0 || i >= RS) continue;
#pragma unroll
                for (int j = 0; j < CW; ++j)
#pragma unroll
                    for (int v = 0; v < 5; ++v) acc[i][j] = fmaf(xv[j + v], k0[u][v], acc[i][j]);
            }
        }
    }
    {   // 3x5 branch
        const int H = p.OH + 2, W = p.OW + 4;
        const float *x = g.x[1] + (long)xs * H * W * g.x_cs[1] + g.x_co[1] + c;
#pragma unroll
        for (int rr = 0; rr < RS + 2; ++rr) {
            const int row = min(i0 + rr, H - 1);
            float xv[CW + 4];
#pragma unroll
            for (int q = 0; q < CW + 4; ++q) xv[q] = x[((long)row * W + min(j0 + q, W - 1)) * g.x_cs[1]];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = rr - u;
                if (i < 0 || i >= RS) continue;
#pragma unroll
                for (int j = 0; j < CW; ++j)
#pragma unroll
                    for (int v = 0; v < 5; ++v) acc[i][j] = fmaf(xv[j + v], k1[u][v], acc[i][j]);
            }
        }
    }
    {   // 5x3 branch
        const int H = p.OH + 4, W = p.OW + 2;
        const float *x = g.x[2] + (long)xs * H * W * g.x_cs[2] + g.x_co[2] + c;
#pragma unroll
        for (int rr = 0; rr < RS + 4; ++rr) {
            const int row = min(i0 + rr, H - 1);
            float xv[CW + 2];
#pragma unroll
            for (int q = 0; q < CW + 2; ++q) xv[q] = x[((long)row * W + min(j0 + q, W - 1)) * g.x_cs[2]];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int i = rr - u;
                if (i < 0 || i >= RS) continue;
#pragma unroll
                for (int j = 0; j < CW; ++j)
#pragma unroll
                    for (int v = 0; v < 3; ++v) acc[i][j] = fmaf(xv[j + v], k2[u][v], acc[i][j]);
            }
        }
    }
    float *o = g.out + ((long)s * p.OH * p.OW) * p.C + c;
#pragma unroll
    for (int i = 0; i < RS; ++i)
#pragma unroll
        for (int j = 0; j < CW; ++j)
            if (i0 + i < p.OH && j0 + j < p.OW) o[((long)(i0 + i) * p.OW + j0 + j) * p.C] = acc[i][j];
}

// Column-thread variant: a thread owns one channel and ONE output column j (all OH rows,
// OH <= 32).  For each of the template's columns v it loads search column j+v once (a run of
// independent loads), then for each template row u streams 25 FMAs against one tap register.
// ~80 VGPRs -> 6 waves per SIMD: latency is hidden by occupancy instead of by a huge
// unrolled register window (the patch variant needs 256 VGPRs and runs 1 wave per SIMD).
template <int HK, int WK>
__device__ __forceinline__ void gdw_col_branch(float (&acc)[32], const float *__restrict__ x, int H, int W, int cs,
                                               const float *__restrict__ z, int zcs, float wsm, int j, int OH)
{
#pragma unroll 1          // one search column live at a time: keeps the kernel at ~80 VGPRs
    for (int v = 0; v < WK; ++v) {
        float col[36];
        const int cidx = min(j + v, W - 1);
#pragma unroll
        for (int r = 0; r < 36; ++r)
            if (r < H) col[r] = x[((long)r * W + cidx) * cs];
#pragma unroll
        for (int u = 0; u < HK; ++u) {
            const float k = wsm * z[(u * WK + v) * zcs];
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (i < OH) acc[i] = fmaf(col[i + u], k, acc[i]);
        }
    }
}

__global__ __launch_bounds__(256) void groupdw_nhwc_col_kernel(const GdwK p)
{
    // blocks: 4 adjacent columns x 64 channels; XCD x%8 -> channel group x%4, sample parity x/4
    const int wave = threadIdx.x >> 6;
    const int ncolg = (p.OW + 3) / 4;
    int cgi, s, colg;
    if (p.C == 256) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        cgi = xcd & 3;
        const int sl = slot / ncolg;
        colg = slot - sl * ncolg;
        s = sl * 2 + (xcd >> 2);
    } else {
        const int ncg = p.C >> 6;
        cgi = blockIdx.x % ncg;
        const int v = blockIdx.x / ncg;
        s = v / ncolg;
        colg = v - s * ncolg;
    }
    const int j = colg * 4 + wave;
    if (s >= p.total || j >= p.OW) return;
    const int c = cgi * 64 + (threadIdx.x & 63);
    int sg = 0;
    while (sg + 1 < p.nseg && s >= p.seg[sg].S) { s -= p.seg[sg].S; ++sg; }
    const GdwSeg &g = p.seg[sg];
    const int xs = s / g.x_rep;
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    {
        const int H = p.OH + 4, W = p.OW + 4;
        gdw_col_branch<5, 5>(acc, g.x[0] + (long)xs * H * W * g.x_cs[0] + g.x_co[0] + c, H, W, g.x_cs[0],
                             g.z[0] + (long)s * 25 * g.z_cs[0] + g.z_co[0] + c, g.z_cs[0], g.wsm[0], j, p.OH);
    }
    {
        const int H = p.OH + 2, W = p.OW + 4;
        gdw_col_branch<3, 5>(acc, g.x[1] + (long)xs * H * W * g.x_cs[1] + g.x_co[1] + c, H, W, g.x_cs[1],
                             g.z[1] + (long)s * 15 * g.z_cs[1] + g.z_co[1] + c, g.z_cs[1], g.wsm[1], j, p.OH);
    }
    {
        const int H = p.OH + 4, W = p.OW + 2;
        gdw_col_branch<5, 3>(acc, g.x[2] + (long)xs * H * W * g.x_cs[2] + g.x_co[2] + c, H, W, g.x_cs[2],
                             g.z[2] + (long)s * 15 * g.z_cs[2] + g.z_co[2] + c, g.z_cs[2], g.wsm[2], j, p.OH);
    }
    float *o = g.out + ((long)s * p.OH * p.OW + j) * p.C + c;
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < p.OH) o[(long)i * p.OW * p.C] = acc[i];
}

// Streaming variant for the bandwidth regime (many samples): one workgroup owns a whole
// (sample, 64-channel group) and walks its search maps top to bottom exactly once.  Row r of
// the three maps is fetched with 16-byte coalesced loads into registers while row r-1 is
// being consumed, parked in a double-buffered LDS row-set, and read back by the compute
// role: wave w owns output columns 5w..5w+4, lane = channel, a 5x5 ring of accumulators holds
// the five output rows a search row contributes to.  Every search element crosses HBM once and
// LDS once per strip that needs it; nothing depends on L1/L2 luck.
template <int NSTRIP>
__global__ __launch_bounds__(64 * NSTRIP) void groupdw_nhwc_stream_kernel(const GdwK p)
{
    constexpr int NT = 64 * NSTRIP;
    extern __shared__ __attribute__((aligned(16))) float rows[];     // [2][npx][64]
    const int W0 = p.OW + 4, W2 = p.OW + 2;
    const int npx = 2 * W0 + W2;                  // pixels of one row-set: x0 | x1 | x2
    int cgi, s;
    if (p.C == 256) {
        const int xcd = blockIdx.x & 7;
        cgi = xcd & 3;
        s = (blockIdx.x >> 3) * 2 + (xcd >> 2);
    } else {
        const int ncg = p.C >> 6;
        cgi = blockIdx.x % ncg;
        s = blockIdx.x / ncg;
    }
    if (s >= p.total) return;
    int sg = 0;
    while (sg + 1 < p.nseg && s >= p.seg[sg].S) { s -= p.seg[sg].S; ++sg; }
    const GdwSeg &g = p.seg[sg];
    const int xs = s / g.x_rep;
    const int tid = threadIdx.x, lane = tid & 63, strip = tid >> 6;
    const int c = cgi * 64 + lane;
    const int j0 = strip * 5;
    const int H0 = p.OH + 4, H1 = p.OH + 2;
    const float *xb0 = g.x[0] + (long)xs * H0 * W0 * g.x_cs[0] + g.x_co[0] + cgi * 64;
    const float *xb1 = g.x[1] + (long)xs * H1 * W0 * g.x_cs[1] + g.x_co[1] + cgi * 64;
    const float *xb2 = g.x[2] + (long)xs * H0 * W2 * g.x_cs[2] + g.x_co[2] + cgi * 64;

    // loader role: chunk q = (pixel, 4-channel group); MAXQ chunks per thread per row-set
    constexpr int MAXQ = 6;                       // (3*27+10)*16 / 320 < 6
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 pre[MAXQ];
    auto fetch = [&](int r) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int q = tid + k * NT;
            const int px = q >> 4, c4 = (q & 15) * 4;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (px < npx) {
                if (px < W0)            v = *(const f4 *)(xb0 + ((long)r * W0 + px) * g.x_cs[0] + c4);
                else if (px < 2 * W0) { if (r < H1) v = *(const f4 *)(xb1 + ((long)r * W0 + px - W0) * g.x_cs[1] + c4); }
                else                    v = *(const f4 *)(xb2 + ((long)r * W2 + px - 2 * W0) * g.x_cs[2] + c4);
            }
            pre[k] = v;
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int q = tid + k * NT;
            const int px = q >> 4, c4 = (q & 15) * 4;
            if (px < npx) *(f4 *)(rows + ((long)buf * npx + px) * 64 + c4) = pre[k];
        }
    };

    float k0[5][5], k1[3][5], k2[5][3];
    {
        const float *z0 = g.z[0] + (long)s * 25 * g.z_cs[0] + g.z_co[0] + c;
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int v = 0; v < 5; ++v) k0[u][v] = g.wsm[0] * z0[(u * 5 + v) * g.z_cs[0]];
        const float *z1 = g.z[1] + (long)s * 15 * g.z_cs[1] + g.z_co[1] + c;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 5; ++v) k1[u][v] = g.wsm[1] * z1[(u * 5 + v) * g.z_cs[1]];
        const float *z2 = g.z[2] + (long)s * 15 * g.z_cs[2] + g.z_co[2] + c;
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) k2[u][v] = g.wsm[2] * z2[(u * 3 + v) * g.z_cs[2]];
    }
    float A[5][5];
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int j = 0; j < 5; ++j) A[u][j] = 0.f;

    fetch(0);
    park(0);
    __syncthreads();
    float *o = g.out + ((long)s * p.OH * p.OW) * p.C + c;
    for (int r = 0; r < H0; ++r) {
        const int buf = r & 1;
        if (r + 1 < H0) fetch(r + 1);
        const float *rs = rows + (long)buf * npx * 64 + lane;
        {
            float xv[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) xv[q] = rs[min(j0 + q, W0 - 1) * 64];
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int v = 0; v < 5; ++v) A[u][j] = fmaf(xv[j + v], k0[u][v], A[u][j]);
        }
        if (r < H1) {
            float xv[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) xv[q] = rs[(W0 + min(j0 + q, W0 - 1)) * 64];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int v = 0; v < 5; ++v) A[u][j] = fmaf(xv[j + v], k1[u][v], A[u][j]);
        }
        {
            float xv[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) xv[q] = rs[(2 * W0 + min(j0 + q, W2 - 1)) * 64];
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int v = 0; v < 3; ++v) A[u][j] = fmaf(xv[j + v], k2[u][v], A[u][j]);
        }
        const int done = r - 4;
        if (done >= 0) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
                if (j0 + j < p.OW) o[((long)done * p.OW + j0 + j) * p.C] = A[4][j];
        }
#pragma unroll
        for (int u = 4; u > 0; --u)
#pragma unroll
            for (int j = 0; j < 5; ++j) A[u][j] = A[u - 1][j];
#pragma unroll
        for (int j = 0; j < 5; ++j) A[0][j] = 0.f;
        if (r + 1 < H0) park(buf ^ 1);
        __syncthreads();
    }
}

// Ring variant for the bandwidth regime: workgroup = one (sample, 64-channel group), wave w =
// output columns 5w..5w+4, lane = channel.  The 55 taps live in LDS ([tap][64], read
// conflict-free one row of 64 lanes at a time) instead of 55 registers per lane; a thread keeps
// only the 5x5 ring of partial output rows and two search rows (current + prefetched), ~90
// VGPRs, so 5+ waves per SIMD hide HBM latency.  Rows stream top to bottom exactly once.
template <int NSTRIP>
__global__ __launch_bounds__(64 * NSTRIP) void groupdw_nhwc_ring_kernel(const GdwK p)
{
    __shared__ float taps[55 * 64];
    int cgi, s;
    if (p.C == 256) {
        const int xcd = blockIdx.x & 7;
        cgi = xcd & 3;
        s = (blockIdx.x >> 3) * 2 + (xcd >> 2);
    } else {
        const int ncg = p.C >> 6;
        cgi = blockIdx.x % ncg;
        s = blockIdx.x / ncg;
    }
    if (s >= p.total) return;
    int sg = 0;
    while (sg + 1 < p.nseg && s >= p.seg[sg].S) { s -= p.seg[sg].S; ++sg; }
    const GdwSeg &g = p.seg[sg];
    const int xs = s / g.x_rep;
    const int lane = threadIdx.x & 63, strip = threadIdx.x >> 6;
    const int c = cgi * 64 + lane;
    const int j0 = strip * 5;
    const int W0 = p.OW + 4, W2 = p.OW + 2, H0 = p.OH + 4, H1 = p.OH + 2;
    for (int t = threadIdx.x; t < 55 * 64; t += 64 * NSTRIP) {
        const int tap = t >> 6, l = t & 63;
        float v;
        if (tap < 25)      v = g.wsm[0] * g.z[0][((long)s * 25 + tap) * g.z_cs[0] + g.z_co[0] + cgi * 64 + l];
        else if (tap < 40) v = g.wsm[1] * g.z[1][((long)s * 15 + tap - 25) * g.z_cs[1] + g.z_co[1] + cgi * 64 + l];
        else               v = g.wsm[2] * g.z[2][((long)s * 15 + tap - 40) * g.z_cs[2] + g.z_co[2] + cgi * 64 + l];
        taps[t] = v;
    }
    __syncthreads();
    const float *x0 = g.x[0] + (long)xs * H0 * W0 * g.x_cs[0] + g.x_co[0] + c;
    const float *x1 = g.x[1] + (long)xs * H1 * W0 * g.x_cs[1] + g.x_co[1] + c;
    const float *x2 = g.x[2] + (long)xs * H0 * W2 * g.x_cs[2] + g.x_co[2] + c;
    // column offsets of this strip (clamped at the map edge; clamped columns feed only
    // outputs that are never stored)
    int o0[9], o2[7];
#pragma unroll
    for (int q = 0; q < 9; ++q) o0[q] = min(j0 + q, W0 - 1);
#pragma unroll
    for (int q = 0; q < 7; ++q) o2[q] = min(j0 + q, W2 - 1);
    float a[9], b[9], d[7], an[9], bn[9], dn[7];
    auto fetch = [&](int r, float (&fa)[9], float (&fb)[9], float (&fd)[7]) {
        const int r1 = min(r, H1 - 1);
#pragma unroll
        for (int q = 0; q < 9; ++q) fa[q] = x0[((long)r * W0 + o0[q]) * g.x_cs[0]];
#pragma unroll
        for (int q = 0; q < 9; ++q) fb[q] = x1[((long)r1 * W0 + o0[q]) * g.x_cs[1]];
#pragma unroll
        for (int q = 0; q < 7; ++q) fd[q] = x2[((long)r * W2 + o2[q]) * g.x_cs[2]];
    };
    float A[5][5];
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int j = 0; j < 5; ++j) A[u][j] = 0.f;
    float *o = g.out + ((long)s * p.OH * p.OW) * p.C + c;
    // (hipcc hoists these loop-invariant LDS reads back into registers — 238 VGPRs; forcing them
    // to stay in LDS with volatile or a register cap made the schedule worse / spilled)
    const float *tp = taps + lane;
    fetch(0, a, b, d);
#pragma unroll 1
    for (int r = 0; r < H0; ++r) {
        fetch(min(r + 1, H0 - 1), an, bn, dn);
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int v = 0; v < 5; ++v) {
                const float k = tp[(u * 5 + v) * 64];
#pragma unroll
                for (int j = 0; j < 5; ++j) A[u][j] = fmaf(a[j + v], k, A[u][j]);
            }
        if (r < H1) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    const float k = tp[(25 + u * 5 + v) * 64];
#pragma unroll
                    for (int j = 0; j < 5; ++j) A[u][j] = fmaf(b[j + v], k, A[u][j]);
                }
        }
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float k = tp[(40 + u * 3 + v) * 64];
#pragma unroll
                for (int j = 0; j < 5; ++j) A[u][j] = fmaf(d[j + v], k, A[u][j]);
            }
        const int done = r - 4;
        if (done >= 0) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
                if (j0 + j < p.OW) o[((long)done * p.OW + j0 + j) * p.C] = A[4][j];
        }
#pragma unroll
        for (int u = 4; u > 0; --u)
#pragma unroll
            for (int j = 0; j < 5; ++j) A[u][j] = A[u - 1][j];
#pragma unroll
        for (int j = 0; j < 5; ++j) A[0][j] = 0.f;
#pragma unroll
        for (int q = 0; q < 9; ++q) { a[q] = an[q]; b[q] = bn[q]; }
#pragma unroll
        for (int q = 0; q < 7; ++q) d[q] = dn[q];
    }
}

// LDS-DMA variant for the bandwidth regime (the auto choice from 64 samples up).
//
// What bounds the register-staged variants above is not HBM: 64-lane dword loads that a wave must
// wait for itself, 5 or 6 strip waves on 4 SIMDs (one SIMD carries twice the work), and 200+
// VGPRs.  Here the search rows reach LDS by `global_load_lds_dwordx4` (1 KiB per wave instruction,
// no VGPRs), issued by loader waves that run two row-sets ahead and do nothing else, so HBM latency
// is covered by bytes in flight, not by occupancy:
//   workgroup = one (sample, 64-channel group) = 4 compute waves + 4 loader waves, 2 per CU;
//   compute wave w = output columns 7w..7w+6 (4 x 7 >= OW: one wave per SIMD, equal work),
//   lane = channel; the 55 taps (pre-scaled by softmax(weight)) and a 5-deep ring of partial output
//   rows live in registers (static ring slots: the row loop is unrolled by 5); every search element
//   is read from LDS once per strip that needs it;
//   LDS: 3 row-set slots x (2 (OW+4) + OW+2) x 256 B = 65.3 KB for OW = 25.
// Four loader waves, not one: a single wave sustains only ~14 GB/s of LDS-DMA however many pieces it
// has in flight (scripts/probes/bw_probe.hip: 256 one-wave blocks x 46 pieces = 3.6 TB/s, 512 = 6.0),
// so one loader per workgroup put a 1.55 us floor under every row.  Each loader owns every fourth
// 1 KiB piece of a row-set (6 per row).
// Loader protocol (one raw s_barrier per row; MI355X guide: "LDS-DMA data is ordered for a ds_read
// only by the issuing wave's counted vmcnt followed by a barrier the reader has passed"): before
// barrier r+1 a loader has issued its pieces of row-set r+2 into the slot row-set r-1 left (every
// compute wave finished reading it before barrier r) and waited vmcnt(6) = only its row-set r+2
// pieces outstanding.  Loaders issue no stores, so their vmcnt counts DMA instructions only, and
// every loader issues exactly 6 per row-set (rows past the 3x5 map's height re-read its last row; the
// loader without a 7th 5x3 piece repeats its last one).
template <int OWT>
struct GdwDma {
    static constexpr int W0 = OWT + 4, W2 = OWT + 2;
    static constexpr int N0 = (W0 + 3) / 4, N2 = (W2 + 3) / 4;     // 1 KiB pieces per branch row
    static constexpr int SLOTF = (2 * W0 + W2) * 64;               // floats per row-set slot: x0 | x1 | x2
    static constexpr int NSLOT = 3;
    static constexpr int NLOAD = 4;                                // loader waves
    static constexpr int PER = 6;                                  // pieces per loader per row-set
    static_assert(N0 == 8 && (N2 == 7 || N2 == 8), "piece schedule below assumes 8 + 8 + 7|8 pieces");
    static constexpr int LDS_BYTES = NSLOT * SLOTF * 4 + 1024;     // + overrun pad of the last strip
};

// NT: non-temporal load policy for data every workgroup reads exactly once (the search rows): MI355X_MICROARCH.md price list,
// 'nt-weights' — issued -> landed 18 % shorter at unchanged issue cost
template <bool NT = false>
__device__ __forceinline__ void gdw_dma16(const float *src, const float *lds_dst)
{
    const uint32_t lds = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds_dst);
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
}

// OT: storage type of the OUTPUT (0 = fp32; 1 = fp16, 2 = bf16: `out` then points to 16-bit elements — the batched mixed-precision
// heads feed these maps to fp16 convolutions, and a separate conversion pass cost 37 + 21 us per batch of 32; same fp32
// arithmetic, one rounding at the store = what that pass produced)
template <int OWT, bool NT = false, int OT = 0>
__global__ __launch_bounds__(512) void groupdw_dma_kernel(const GdwK p)
{
    using G = GdwDma<OWT>;
    constexpr int W0 = G::W0, W2 = G::W2, SLOTF = G::SLOTF;
    extern __shared__ __attribute__((aligned(16))) float rows[];     // [NSLOT][SLOTF]
    int cgi, s;
    if (p.C == 256) {
        const int xcd = blockIdx.x & 7;
        cgi = xcd & 3;
        s = (blockIdx.x >> 3) * 2 + (xcd >> 2);
    } else {
        const int ncg = p.C >> 6;
        cgi = blockIdx.x % ncg;
        s = blockIdx.x / ncg;
    }
    if (s >= p.total) return;
    int sg = 0;
    while (sg + 1 < p.nseg && s >= p.seg[sg].S) { s -= p.seg[sg].S; ++sg; }
    const GdwSeg &g = p.seg[sg];
    const int xs = s / g.x_rep;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H0 = p.OH + 4, H1 = p.OH + 2;

    if (wave >= 4) {
        // ---------------- loader waves: lane = (pixel of a 4-pixel piece, 16-byte channel chunk) ----------------
        const int lw = wave - 4;
        const int sub = lane >> 4, c4 = (lane & 15) * 4;
        // this loader's six pieces: k = lw, lw + 4 of the 5x5 row, of the 3x5 row, of the 5x3 row (the
        // last loader's second 5x3 piece does not exist when the row has 7: it repeats piece lw)
        const int ka = lw, kb = lw + 4;
        const int kd = kb < G::N2 ? kb : ka;
        const float *x0 = g.x[0] + (long)xs * H0 * W0 * g.x_cs[0] + g.x_co[0] + cgi * 64 + c4;
        const float *x1 = g.x[1] + (long)xs * H1 * W0 * g.x_cs[1] + g.x_co[1] + cgi * 64 + c4;
        const float *x2 = g.x[2] + (long)xs * H0 * W2 * g.x_cs[2] + g.x_co[2] + cgi * 64 + c4;
        // pixels of a piece beyond the row end are not fetched (exec-masked lanes write nothing)
        const bool oa0 = ka * 4 + sub < W0, ob0 = kb * 4 + sub < W0, oa2 = ka * 4 + sub < W2, od2 = kd * 4 + sub < W2;
        const float *pa0 = x0 + (long)(ka * 4 + sub) * g.x_cs[0], *pb0 = x0 + (long)(kb * 4 + sub) * g.x_cs[0];
        const float *pa1 = x1 + (long)(ka * 4 + sub) * g.x_cs[1], *pb1 = x1 + (long)(kb * 4 + sub) * g.x_cs[1];
        const float *pa2 = x2 + (long)(ka * 4 + sub) * g.x_cs[2], *pd2 = x2 + (long)(kd * 4 + sub) * g.x_cs[2];
        const long rs0 = (long)W0 * g.x_cs[0], rs1 = (long)W0 * g.x_cs[1], rs2 = (long)W2 * g.x_cs[2];
        auto issue = [&](int r, int slot) {
            float *dst = rows + slot * SLOTF;
            const long o0 = r * rs0, o1 = min(r, H1 - 1) * rs1, o2 = r * rs2;
            if (oa0) gdw_dma16<NT>(pa0 + o0, dst + ka * 256);
            if (ob0) gdw_dma16<NT>(pb0 + o0, dst + kb * 256);
            if (oa0) gdw_dma16<NT>(pa1 + o1, dst + W0 * 64 + ka * 256);
            if (ob0) gdw_dma16<NT>(pb1 + o1, dst + W0 * 64 + kb * 256);
            if (oa2) gdw_dma16<NT>(pa2 + o2, dst + 2 * W0 * 64 + ka * 256);
            if (od2) gdw_dma16<NT>(pd2 + o2, dst + 2 * W0 * 64 + kd * 256);
        };
        issue(0, 0);
        if (H0 > 1) {
            issue(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(G::PER) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        int slot2 = 2;                                    // slot of row-set r + 2
        for (int r = 0; r < H0; ++r) {
            if (r + 2 < H0) {
                issue(r + 2, slot2);
                asm volatile("s_waitcnt vmcnt(%0)" :: "i"(G::PER) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            slot2 = slot2 == 2 ? 0 : slot2 + 1;
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---------------- compute waves ----------------
    const int j0 = wave * 7;
    const int c = cgi * 64 + lane;
    float k0[25], k1[15], k2[15];                        // taps while the first row-sets are in flight
    {
        const float *z0 = g.z[0] + (long)s * 25 * g.z_cs[0] + g.z_co[0] + c;
        const float *z1 = g.z[1] + (long)s * 15 * g.z_cs[1] + g.z_co[1] + c;
        const float *z2 = g.z[2] + (long)s * 15 * g.z_cs[2] + g.z_co[2] + c;
#pragma unroll
        for (int t = 0; t < 25; ++t) k0[t] = g.wsm[0] * z0[t * g.z_cs[0]];
#pragma unroll
        for (int t = 0; t < 15; ++t) k1[t] = g.wsm[1] * z1[t * g.z_cs[1]];
#pragma unroll
        for (int t = 0; t < 15; ++t) k2[t] = g.wsm[2] * z2[t * g.z_cs[2]];
    }
    float A[5][7];
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int j = 0; j < 7; ++j) A[u][j] = 0.f;
    float *o = g.out + ((long)s * p.OH * p.OW + j0) * p.C + c;
    const int nvalid = min(7, p.OW - j0);               // columns of this strip that exist
    const float *strip = rows + j0 * 64 + lane;        // + slot * SLOTF + pixel * 64

    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    auto step = [&](auto rho_c, int r, int slot) {
        constexpr int RHO = decltype(rho_c)::value;      // r mod 5: ring slot of output row r
        const float *rs = strip + slot * SLOTF;
        {   // 5x5 branch: search row r feeds output rows r-u
            float xv[11];
#pragma unroll
            for (int q = 0; q < 11; ++q) xv[q] = rs[q * 64];
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int v = 0; v < 5; ++v)
#pragma unroll
                    for (int j = 0; j < 7; ++j) A[(RHO - u + 5) % 5][j] = fmaf(xv[j + v], k0[u * 5 + v], A[(RHO - u + 5) % 5][j]);
        }
        if (r < H1) {   // 3x5 branch
            float xv[11];
#pragma unroll
            for (int q = 0; q < 11; ++q) xv[q] = rs[(W0 + q) * 64];
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 5; ++v)
#pragma unroll
                    for (int j = 0; j < 7; ++j) A[(RHO - u + 5) % 5][j] = fmaf(xv[j + v], k1[u * 5 + v], A[(RHO - u + 5) % 5][j]);
        }
        {   // 5x3 branch
            float xv[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) xv[q] = rs[(2 * W0 + q) * 64];
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v)
#pragma unroll
                    for (int j = 0; j < 7; ++j) A[(RHO - u + 5) % 5][j] = fmaf(xv[j + v], k2[u * 3 + v], A[(RHO - u + 5) % 5][j]);
        }
        // output row r-4 is complete; its ring slot is (RHO + 1) % 5 and becomes output row r+1
        constexpr int DONE = (RHO + 1) % 5;
        if (r >= 4) {
            if constexpr (OT == 0) {
                float *orow = o + (long)(r - 4) * p.OW * p.C;
#pragma unroll
                for (int j = 0; j < 7; ++j)
                    if (j < nvalid) orow[(long)j * p.C] = A[DONE][j];
            } else {
                uint16_t *orow = (uint16_t *)g.out + (o - g.out) + (long)(r - 4) * p.OW * p.C;
#pragma unroll
                for (int j = 0; j < 7; ++j)
                    if (j < nvalid)
                        orow[(long)j * p.C] = OT == 1 ? __builtin_bit_cast(uint16_t, (_Float16)A[DONE][j]) : __builtin_bit_cast(uint16_t, (__bf16)A[DONE][j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) A[DONE][j] = 0.f;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    int slot = 0;
    auto next = [&]() { const int cur = slot; slot = slot == 2 ? 0 : slot + 1; return cur; };
    for (int r = 0; r < H0; r += 5) {
        step(I0{}, r, next());
        if (r + 1 < H0) step(I1{}, r + 1, next());
        if (r + 2 < H0) step(I2{}, r + 2, next());
        if (r + 3 < H0) step(I3{}, r + 3, next());
        if (r + 4 < H0) step(I4{}, r + 4, next());
    }
}

// -------------------------------------------------------------------------------------
// (1c) the LDS-DMA kernel, PERSISTENT (round 4).  groupdw_dma_kernel pays, per (sample, channel group) unit of ~80 us, a start-up
// in which nothing streams — 55 strided tap loads per compute lane (one HBM round trip), the first two row-sets (another) — and
// a workgroup turn-over (the dispatcher refills a slot only after its workgroup has drained): at 16 rounds of 512 workgroups
// that is ~5 % of the launch during which a slot's share of the HBM queue is empty.  Here the grid is the resident set
// (2 workgroups per CU) and a workgroup walks its units as ONE flat row-set sequence: while the compute waves finish rows
// H0-2 and H0-1 of a unit the loaders are already fetching row-sets 0 and 1 of the next one, and its taps: the loaders also
// DMA the next unit's 55 x 64 taps into an LDS tap image (15 KiB) during the current unit, so the compute waves change units
// with 55 LDS reads instead of a global round trip.  Same arithmetic in the same order as groupdw_dma_kernel: bit-identical.
// LDS per workgroup: 3 row-set slots (65 280 B) + 1 KiB overrun pad + tap image 60 rows x 256 B = 81 664 B -> two per CU.
template <int OWT>
struct GdwDmaP {
    using G = GdwDma<OWT>;
    static constexpr int TAP_ROWS = 28 + 16 + 16;                   // 25 | 15 | 15 taps in whole 4-tap DMA pieces
    static constexpr int TAP_OFF = G::NSLOT * G::SLOTF + 256;       // floats: behind the ring and its overrun pad
    static constexpr int LDS_BYTES = (TAP_OFF + TAP_ROWS * 64) * 4;
};

template <int OWT, bool NT>
__global__ __launch_bounds__(512, 4) void groupdw_dmap_kernel(const GdwK p, int nunits)
{
    using G = GdwDma<OWT>;
    using GP = GdwDmaP<OWT>;
    constexpr int W0 = G::W0, W2 = G::W2, SLOTF = G::SLOTF;
    extern __shared__ __attribute__((aligned(16))) float rows[];     // [NSLOT][SLOTF] | pad | taps[60][64]
    float *taps = rows + GP::TAP_OFF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H0 = p.OH + 4, H1 = p.OH + 2;
    const int ncg = p.C >> 6;

    // unit -> (segment, sample within the segment, channel group); false when the unit does not exist (odd sample count)
    struct Unit { int sg, s, cgi; };
    auto decode = [&](int u, Unit &o) -> bool {
        int cgi, s;
        if (p.C == 256) {
            const int x8 = u & 7;
            cgi = x8 & 3;
            s = (u >> 3) * 2 + (x8 >> 2);
        } else {
            cgi = u % ncg;
            s = u / ncg;
        }
        if (s >= p.total) return false;
        int sg = 0;
        while (sg + 1 < p.nseg && s >= p.seg[sg].S) { s -= p.seg[sg].S; ++sg; }
        o.sg = sg; o.s = s; o.cgi = cgi;
        return true;
    };
    auto next_valid = [&](int u, Unit &o) -> int {                   // first existing unit at or after u in this workgroup's walk
        while (u < nunits && !decode(u, o)) u += gridDim.x;
        return u;
    };

    if (wave >= 4) {
        // ---------------- loader waves ----------------
        const int lw = wave - 4;
        const int sub = lane >> 4, c4 = (lane & 15) * 4;
        const int ka = lw, kb = lw + 4;
        const int kd = kb < G::N2 ? kb : ka;
        const bool oa0 = ka * 4 + sub < W0, ob0 = kb * 4 + sub < W0, oa2 = ka * 4 + sub < W2, od2 = kd * 4 + sub < W2;
        // row-set r of unit v into `slot`
        auto issue_rows = [&](const Unit &v, int r, int slot) {
            const GdwSeg &g = p.seg[v.sg];
            const int xs = v.s / g.x_rep;
            const float *x0 = g.x[0] + (long)xs * H0 * W0 * g.x_cs[0] + g.x_co[0] + v.cgi * 64 + c4;
            const float *x1 = g.x[1] + (long)xs * H1 * W0 * g.x_cs[1] + g.x_co[1] + v.cgi * 64 + c4;
            const float *x2 = g.x[2] + (long)xs * H0 * W2 * g.x_cs[2] + g.x_co[2] + v.cgi * 64 + c4;
            const long o0 = ((long)r * W0) * g.x_cs[0], o1 = ((long)min(r, H1 - 1) * W0) * g.x_cs[1], o2 = ((long)r * W2) * g.x_cs[2];
            float *dst = rows + slot * SLOTF;
            if (oa0) gdw_dma16<NT>(x0 + o0 + (long)(ka * 4 + sub) * g.x_cs[0], dst + ka * 256);
            if (ob0) gdw_dma16<NT>(x0 + o0 + (long)(kb * 4 + sub) * g.x_cs[0], dst + kb * 256);
            if (oa0) gdw_dma16<NT>(x1 + o1 + (long)(ka * 4 + sub) * g.x_cs[1], dst + W0 * 64 + ka * 256);
            if (ob0) gdw_dma16<NT>(x1 + o1 + (long)(kb * 4 + sub) * g.x_cs[1], dst + W0 * 64 + kb * 256);
            if (oa2) gdw_dma16<NT>(x2 + o2 + (long)(ka * 4 + sub) * g.x_cs[2], dst + 2 * W0 * 64 + ka * 256);
            if (od2) gdw_dma16<NT>(x2 + o2 + (long)(kd * 4 + sub) * g.x_cs[2], dst + 2 * W0 * 64 + kd * 256);
        };
        // the unit's taps -> LDS tap image: rows 0..24 (5x5), 28..42 (3x5), 44..58 (5x3); a piece = 4 taps x 64 channels.
        // 7 + 4 + 4 pieces over four loaders.  Always issued BEFORE a row-set's pieces: the wait behind those
        // (vmcnt(PER): only the newest PER operations outstanding, in-order retirement) then covers the taps too.
        auto issue_taps = [&](const Unit &v) {
            const GdwSeg &g = p.seg[v.sg];
            const float *z0 = g.z[0] + (long)v.s * 25 * g.z_cs[0] + g.z_co[0] + v.cgi * 64 + c4;
            const float *z1 = g.z[1] + (long)v.s * 15 * g.z_cs[1] + g.z_co[1] + v.cgi * 64 + c4;
            const float *z2 = g.z[2] + (long)v.s * 15 * g.z_cs[2] + g.z_co[2] + v.cgi * 64 + c4;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int pc = lw + 4 * k;                            // 5x5 pieces 0..6
                if (pc < 7 && pc * 4 + sub < 25) gdw_dma16<false>(z0 + (long)(pc * 4 + sub) * g.z_cs[0], taps + pc * 256);
            }
            if (lw * 4 + sub < 15) {
                gdw_dma16<false>(z1 + (long)(lw * 4 + sub) * g.z_cs[1], taps + (28 + lw * 4) * 64);
                gdw_dma16<false>(z2 + (long)(lw * 4 + sub) * g.z_cs[2], taps + (44 + lw * 4) * 64);
            }
        };
        Unit u, un;
        int ui = next_valid(blockIdx.x, u);
        if (ui >= nunits) return;
        issue_taps(u);
        issue_rows(u, 0, 0);
        issue_rows(u, 1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(G::PER) : "memory");
        __builtin_amdgcn_s_barrier();
        int slot2 = 2;
        while (ui < nunits) {
            const int uni = next_valid(ui + gridDim.x, un);
            const bool more_units = uni < nunits;
            for (int r = 0; r < H0; ++r) {
                if (r == 1 && more_units) issue_taps(un);           // every compute wave read this unit's taps before barrier (u, 0)
                bool more = true;
                if (r + 2 < H0) issue_rows(u, r + 2, slot2);
                else if (more_units) issue_rows(un, r + 2 - H0, slot2);
                else more = false;
                if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(G::PER) : "memory");
                else      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                slot2 = slot2 == 2 ? 0 : slot2 + 1;
                __builtin_amdgcn_s_barrier();
            }
            ui = uni;
            u = un;
        }
        return;
    }

    // ---------------- compute waves ----------------
    const int j0 = wave * 7;
    const int nvalid = min(7, p.OW - j0);
    const float *strip = rows + j0 * 64 + lane;
    Unit u;
    int ui = next_valid(blockIdx.x, u);
    if (ui >= nunits) return;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int slot = 0;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    while (ui < nunits) {
        const GdwSeg &g = p.seg[u.sg];
        float k0[25], k1[15], k2[15];
#pragma unroll
        for (int t = 0; t < 25; ++t) k0[t] = g.wsm[0] * taps[t * 64 + lane];
#pragma unroll
        for (int t = 0; t < 15; ++t) k1[t] = g.wsm[1] * taps[(28 + t) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 15; ++t) k2[t] = g.wsm[2] * taps[(44 + t) * 64 + lane];
        float A[5][7];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int j = 0; j < 7; ++j) A[a][j] = 0.f;
        float *o = g.out + ((long)u.s * p.OH * p.OW + j0) * p.C + u.cgi * 64 + lane;
        auto step = [&](auto rho_c, int r, int sl) {
            constexpr int RHO = decltype(rho_c)::value;
            const float *rs = strip + sl * SLOTF;
            {
                float xv[11];
#pragma unroll
                for (int q = 0; q < 11; ++q) xv[q] = rs[q * 64];
#pragma unroll
                for (int a = 0; a < 5; ++a)
#pragma unroll
                    for (int v = 0; v < 5; ++v)
#pragma unroll
                        for (int j = 0; j < 7; ++j) A[(RHO - a + 5) % 5][j] = fmaf(xv[j + v], k0[a * 5 + v], A[(RHO - a + 5) % 5][j]);
            }
            if (r < H1) {
                float xv[11];
#pragma unroll
                for (int q = 0; q < 11; ++q) xv[q] = rs[(W0 + q) * 64];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int v = 0; v < 5; ++v)
#pragma unroll
                        for (int j = 0; j < 7; ++j) A[(RHO - a + 5) % 5][j] = fmaf(xv[j + v], k1[a * 5 + v], A[(RHO - a + 5) % 5][j]);
            }
            {
                float xv[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) xv[q] = rs[(2 * W0 + q) * 64];
#pragma unroll
                for (int a = 0; a < 5; ++a)
#pragma unroll
                    for (int v = 0; v < 3; ++v)
#pragma unroll
                        for (int j = 0; j < 7; ++j) A[(RHO - a + 5) % 5][j] = fmaf(xv[j + v], k2[a * 3 + v], A[(RHO - a + 5) % 5][j]);
            }
            constexpr int DONE = (RHO + 1) % 5;
            if (r >= 4) {
                float *orow = o + (long)(r - 4) * p.OW * p.C;
#pragma unroll
                for (int j = 0; j < 7; ++j)
                    if (j < nvalid) orow[(long)j * p.C] = A[DONE][j];
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) A[DONE][j] = 0.f;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto next = [&]() { const int cur = slot; slot = slot == 2 ? 0 : slot + 1; return cur; };
        for (int r = 0; r < H0; r += 5) {
            step(I0{}, r, next());
            if (r + 1 < H0) step(I1{}, r + 1, next());
            if (r + 2 < H0) step(I2{}, r + 2, next());
            if (r + 3 < H0) step(I3{}, r + 3, next());
            if (r + 4 < H0) step(I4{}, r + 4, next());
        }
        ui = next_valid(ui + gridDim.x, u);
    }
}

// -------------------------------------------------------------------------------------
// (2) per-plane xcorr on NCHW
// -------------------------------------------------------------------------------------
template <int HK, int WK>
__global__ __launch_bounds__(256) void xcorr_planes_kernel(
    const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out,
    int P, int Hx, int Wx, int per_wave)
{
    __shared__ float tz[4][2][HK * WK];          // template tiles of this block's waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int OHh = Hx - HK + 1, OWw = Wx - WK + 1;
    const int half = per_wave == 2 ? (lane >> 5) : 0;
    const int j = per_wave == 2 ? (lane & 31) : lane;
    const int nunits = (P + per_wave - 1) / per_wave;
    for (int unit = blockIdx.x * 4 + wave; unit < nunits; unit += gridDim.x * 4) {
        const int plane = unit * per_wave + half;
        const bool live = plane < P;
        // stage the template tile(s) in LDS, then every lane reads taps by broadcast
        for (int t = lane; t < per_wave * HK * WK; t += 64) {
            const int h = t / (HK * WK), e = t - h * HK * WK;
            const int pl = unit * per_wave + h;
            tz[wave][h][e] = pl < P ? k[(long)pl * HK * WK + e] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        float kt[HK][WK];
#pragma unroll
        for (int u = 0; u < HK; ++u)
#pragma unroll
            for (int v = 0; v < WK; ++v) kt[u][v] = tz[wave][half][u * WK + v];

        const float *xp = x + (long)(live ? plane : 0) * Hx * Wx;
        float *op = out + (long)(live ? plane : 0) * OHh * OWw;
        // win[u][v] = x[row r-HK+1+u][j+v]; horizontally shifted copies come from lane j+v
        float win[HK][WK];
#pragma unroll
        for (int u = 0; u < HK; ++u)
#pragma unroll
            for (int v = 0; v < WK; ++v) win[u][v] = 0.f;
        for (int r = 0; r < Hx; ++r) {
#pragma unroll
            for (int u = 0; u < HK - 1; ++u)
#pragma unroll
                for (int v = 0; v < WK; ++v) win[u][v] = win[u + 1][v];
            const float xv = (live && j < Wx) ? xp[(long)r * Wx + j] : 0.f;
            win[HK - 1][0] = xv;
#pragma unroll
            for (int v = 1; v < WK; ++v) win[HK - 1][v] = __shfl_down(xv, v, 64);
            const int i = r - (HK - 1);
            if (i >= 0) {
                float acc = 0.f;
#pragma unroll
                for (int u = 0; u < HK; ++u)
#pragma unroll
                    for (int v = 0; v < WK; ++v) acc = fmaf(win[u][v], kt[u][v], acc);
                if (live && j < OWw) op[(long)i * OWw + j] = acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// any template size: one thread per output element (slow path, correctness only)
__global__ __launch_bounds__(256) void xcorr_planes_generic(
    const float *__restrict__ x, const float *__restrict__ k, float *__restrict__ out,
    int P, int Hx, int Wx, int Hk, int Wk)
{
    const int OHh = Hx - Hk + 1, OWw = Wx - Wk + 1;
    const long total = (long)P * OHh * OWw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int jx = (int)(idx % OWw);
        const int iy = (int)((idx / OWw) % OHh);
        const long pl = idx / ((long)OWw * OHh);
        float acc = 0.f;
        for (int u = 0; u < Hk; ++u)
            for (int v = 0; v < Wk; ++v)
                acc = fmaf(x[(pl * Hx + iy + u) * Wx + jx + v], k[(pl * Hk + u) * Wk + v], acc);
        out[idx] = acc;
    }
}

template <int HK, int WK>
int launch_planes(hipStream_t s, const float *x, const float *k, float *out, int P, int Hx, int Wx)
{
    const int per_wave = Wx <= 32 ? 2 : 1;
    const int nunits = (P + per_wave - 1) / per_wave;
    int blocks = (nunits + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((xcorr_planes_kernel<HK, WK>), dim3(blocks), dim3(256), 0, s, x, k, out, P, Hx, Wx, per_wave);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

}  // namespace

extern "C" int usot_xcorr_depthwise_f32(void *stream, const float *x, const float *k, float *out,
                                        int P, int Hx, int Wx, int Hk, int Wk)
{
    if (!x || !k || !out || P < 0 || Hk < 1 || Wk < 1 || Hx < Hk || Wx < Wk) return USOT_EINVAL;
    if (P == 0) return USOT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (Wx <= 64) {
        if (Hk == 5 && Wk == 5) return launch_planes<5, 5>(s, x, k, out, P, Hx, Wx);
        if (Hk == 3 && Wk == 5) return launch_planes<3, 5>(s, x, k, out, P, Hx, Wx);
        if (Hk == 5 && Wk == 3) return launch_planes<5, 3>(s, x, k, out, P, Hx, Wx);
    }
    const long total = (long)P * (Hx - Hk + 1) * (Wx - Wk + 1);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(xcorr_planes_generic, dim3(blocks), dim3(256), 0, s, x, k, out, P, Hx, Wx, Hk, Wk);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

// variant the launcher picks for `total` samples of an OW-wide response when cols_per_thread == 0
static int groupdw_auto_mode(int total, int OW)
{
    if (total >= 64 && (OW == 25 || OW == 27)) return 6;      // LDS-DMA row streaming
    const int nstrip0 = (OW + 4) / 5;
    return (total >= 64 && (nstrip0 == 5 || nstrip0 == 6)) ? 4 : 1;
}

extern "C" int usot_groupdw_auto_variant(int total_samples, int OW)
{
    return groupdw_auto_mode(total_samples, OW);
}

static int groupdw_multi_impl(void *stream, const usot_groupdw_desc *d, int nseg, int out_dtype)
{
    const int usot_dv = usot_device_slot();        // per-device launcher state below (common.h)
    if (usot_dv < 0) return USOT_ESTATE;
    if (!d || nseg < 1 || nseg > 3) return USOT_EINVAL;
    static const int hk[3] = {5, 3, 5}, wk[3] = {5, 5, 3};
    GdwK p;
    p.nseg = nseg; p.OH = d[0].OH; p.OW = d[0].OW; p.C = d[0].C;
    if (p.OH < 1 || p.OW < 1 || p.C <= 0 || (p.C & 63)) return USOT_EINVAL;
    int total = 0;
    for (int sidx = 0; sidx < nseg; ++sidx) {
        const usot_groupdw_desc &q = d[sidx];
        if (!q.out || q.S <= 0 || q.x_rep < 1 || q.OH != p.OH || q.OW != p.OW || q.C != p.C) return USOT_EINVAL;
        GdwSeg &g = p.seg[sidx];
        for (int b = 0; b < 3; ++b) {
            if (!q.x[b] || !q.z[b] || q.hk[b] != hk[b] || q.wk[b] != wk[b]) return USOT_EINVAL;
            g.x[b] = q.x[b]; g.z[b] = q.z[b];
            g.x_cs[b] = q.x_cs[b] > 0 ? q.x_cs[b] : q.C; g.x_co[b] = q.x_co[b];
            g.z_cs[b] = q.z_cs[b] > 0 ? q.z_cs[b] : q.C; g.z_co[b] = q.z_co[b];
            g.wsm[b] = q.wsm[b];
        }
        g.out = q.out; g.S = q.S; g.x_rep = q.x_rep;
        total += q.S;
    }
    // variant: 0 -> auto (5x1 strips for a frame's 9 samples, ring for >= 64 samples);
    // 1 strips, 5/50/52 5x5 patches (register budgets), 2 column threads, 3 LDS row streaming, 4 ring
    int mode = d[0].cols_per_thread != 0 ? d[0].cols_per_thread : groupdw_auto_mode(total, p.OW);
    if (out_dtype) {                     // 16-bit outputs: the LDS-DMA kernel only (25- / 27-wide responses)
        if (out_dtype != 1 && out_dtype != 2) return USOT_EINVAL;
        if (p.OW != 25 && p.OW != 27) return USOT_EINVAL;
        mode = 6;
    }
    if (d[0].cols_per_thread == 0 && mode == 6 && !out_dtype) {
        // search rows that ONE sample reads once stream with the non-temporal policy (round 4, scripts/xcorr_probe2.py: 128 samples
        // 90 -> 82 us, 512: 336 -> 332, 2048: equal; MI355X_MICROARCH.md 'nt-weights'); maps shared by x_rep samples (the memory
        // branch of a frame batch) keep the default policy: their re-reads are L2 hits
        bool once = true;
        for (int sidx = 0; sidx < nseg; ++sidx) once = once && d[sidx].x_rep == 1;
        if (once) mode = 7;
    }
    hipStream_t s = (hipStream_t)stream;
    if (mode != 0 && mode != 1 && mode != 2 && mode != 3 && mode != 4 && mode != 5 && mode != 6 && mode != 50 && mode != 52 && mode != 7 && mode != 8 && mode != 9) return USOT_EINVAL;
    if (mode == 8 || mode == 9) {   // persistent LDS-DMA: the resident set of workgroups walks the (sample, channel group) units
        if (p.OW != 25 && p.OW != 27) return USOT_EINVAL;
        for (int sidx = 0; sidx < nseg; ++sidx)
            for (int b = 0; b < 3; ++b)        // 16-byte DMA pieces of search rows AND taps
                if (((uintptr_t)p.seg[sidx].x[b] & 15) || (p.seg[sidx].x_cs[b] & 3) || (p.seg[sidx].x_co[b] & 3) ||
                    ((uintptr_t)p.seg[sidx].z[b] & 15) || (p.seg[sidx].z_cs[b] & 3) || (p.seg[sidx].z_co[b] & 3)) return USOT_EINVAL;
        p.total = total;
        p.nty = p.ntx = 1;
        const long nunits = p.C == 256 ? 8L * ((total + 1) / 2) : (long)(p.C / 64) * total;
        if (nunits > 0x7fffffffL) return USOT_EINVAL;
        static int slots_d[USOT_MAX_DEV] = {};
    int &slots = slots_d[usot_dv];
        if (!slots) {
            (void)hipFuncSetAttribute((const void *)groupdw_dmap_kernel<25, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDmaP<25>::LDS_BYTES);
            (void)hipFuncSetAttribute((const void *)groupdw_dmap_kernel<25, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDmaP<25>::LDS_BYTES);
            (void)hipFuncSetAttribute((const void *)groupdw_dmap_kernel<27, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDmaP<27>::LDS_BYTES);
            (void)hipFuncSetAttribute((const void *)groupdw_dmap_kernel<27, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDmaP<27>::LDS_BYTES);
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            slots = 2 * cus;                   // two workgroups per CU (LDS: 2 x 81 664 B of 160 KiB; 128 VGPRs x 16 waves)
            slots -= slots % 8;                // a multiple of 8 keeps a workgroup's units on one XCD / channel-group parity
        }
        const int grid = (int)(nunits < slots ? nunits : slots);
        const size_t lds = p.OW == 25 ? GdwDmaP<25>::LDS_BYTES : GdwDmaP<27>::LDS_BYTES;
        if (p.OW == 25) {
            if (mode == 9) hipLaunchKernelGGL((groupdw_dmap_kernel<25, true>), dim3(grid), dim3(512), lds, s, p, (int)nunits);
            else           hipLaunchKernelGGL((groupdw_dmap_kernel<25, false>), dim3(grid), dim3(512), lds, s, p, (int)nunits);
        } else {
            if (mode == 9) hipLaunchKernelGGL((groupdw_dmap_kernel<27, true>), dim3(grid), dim3(512), lds, s, p, (int)nunits);
            else           hipLaunchKernelGGL((groupdw_dmap_kernel<27, false>), dim3(grid), dim3(512), lds, s, p, (int)nunits);
        }
        USOT_CHECK_LAUNCH();
        return USOT_OK;
    }
    if (mode == 6 || mode == 7) {            // LDS-DMA: workgroup per (sample, 64-channel group), loader wave + 4 strip waves
        if (p.OW != 25 && p.OW != 27) return USOT_EINVAL;
        for (int sidx = 0; sidx < nseg; ++sidx)
            for (int b = 0; b < 3; ++b)        // 16-byte DMA pieces
                if (((uintptr_t)p.seg[sidx].x[b] & 15) || (p.seg[sidx].x_cs[b] & 3) || (p.seg[sidx].x_co[b] & 3)) return USOT_EINVAL;
        p.total = total;
        p.nty = p.ntx = 1;
        const long nb = p.C == 256 ? 8L * ((total + 1) / 2) : (long)(p.C / 64) * total;
        static bool attr_set_d[USOT_MAX_DEV] = {};
    bool &attr_set = attr_set_d[usot_dv];
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)groupdw_dma_kernel<25, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDma<25>::LDS_BYTES);
            (void)hipFuncSetAttribute((const void *)groupdw_dma_kernel<27, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDma<27>::LDS_BYTES);
            (void)hipFuncSetAttribute((const void *)groupdw_dma_kernel<25, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDma<25>::LDS_BYTES);
            (void)hipFuncSetAttribute((const void *)groupdw_dma_kernel<27, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDma<27>::LDS_BYTES);
            attr_set = true;
        }
        if (out_dtype) {
            static bool attr_lp_d[USOT_MAX_DEV] = {};
    bool &attr_lp = attr_lp_d[usot_dv];
            if (!attr_lp) {
                (void)hipFuncSetAttribute((const void *)groupdw_dma_kernel<25, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDma<25>::LDS_BYTES);
                (void)hipFuncSetAttribute((const void *)groupdw_dma_kernel<27, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDma<27>::LDS_BYTES);
                (void)hipFuncSetAttribute((const void *)groupdw_dma_kernel<25, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDma<25>::LDS_BYTES);
                (void)hipFuncSetAttribute((const void *)groupdw_dma_kernel<27, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GdwDma<27>::LDS_BYTES);
                attr_lp = true;
            }
            if (p.OW == 25) {
                if (out_dtype == 1) hipLaunchKernelGGL((groupdw_dma_kernel<25, false, 1>), dim3((unsigned)nb), dim3(512), GdwDma<25>::LDS_BYTES, s, p);
                else                hipLaunchKernelGGL((groupdw_dma_kernel<25, false, 2>), dim3((unsigned)nb), dim3(512), GdwDma<25>::LDS_BYTES, s, p);
            } else {
                if (out_dtype == 1) hipLaunchKernelGGL((groupdw_dma_kernel<27, false, 1>), dim3((unsigned)nb), dim3(512), GdwDma<27>::LDS_BYTES, s, p);
                else                hipLaunchKernelGGL((groupdw_dma_kernel<27, false, 2>), dim3((unsigned)nb), dim3(512), GdwDma<27>::LDS_BYTES, s, p);
            }
            USOT_CHECK_LAUNCH();
            return USOT_OK;
        }
        if (p.OW == 25) {
            if (mode == 7) hipLaunchKernelGGL((groupdw_dma_kernel<25, true>), dim3((unsigned)nb), dim3(512), GdwDma<25>::LDS_BYTES, s, p);
            else           hipLaunchKernelGGL((groupdw_dma_kernel<25, false>), dim3((unsigned)nb), dim3(512), GdwDma<25>::LDS_BYTES, s, p);
        } else {
            if (mode == 7) hipLaunchKernelGGL((groupdw_dma_kernel<27, true>), dim3((unsigned)nb), dim3(512), GdwDma<27>::LDS_BYTES, s, p);
            else           hipLaunchKernelGGL((groupdw_dma_kernel<27, false>), dim3((unsigned)nb), dim3(512), GdwDma<27>::LDS_BYTES, s, p);
        }
        USOT_CHECK_LAUNCH();
        return USOT_OK;
    }
    if (mode == 4) {            // ring: workgroup per (sample, 64-channel group), taps in LDS
        const int nstrip = (p.OW + 4) / 5;
        if (nstrip != 5 && nstrip != 6) return USOT_EINVAL;
        p.total = total;
        p.nty = p.ntx = 1;
        const long nb = p.C == 256 ? 8L * ((total + 1) / 2) : (long)(p.C / 64) * total;
        if (nstrip == 5) hipLaunchKernelGGL(groupdw_nhwc_ring_kernel<5>, dim3((unsigned)nb), dim3(320), 0, (hipStream_t)stream, p);
        else             hipLaunchKernelGGL(groupdw_nhwc_ring_kernel<6>, dim3((unsigned)nb), dim3(384), 0, (hipStream_t)stream, p);
        USOT_CHECK_LAUNCH();
        return USOT_OK;
    }
    if (mode == 3) {            // streaming: one workgroup per (sample, 64-channel group)
        const int nstrip = (p.OW + 4) / 5;
        if (nstrip != 5 && nstrip != 6) return USOT_EINVAL;
        p.total = total;
        p.nty = p.ntx = 1;
        const long nb = p.C == 256 ? 8L * ((total + 1) / 2) : (long)(p.C / 64) * total;
        const size_t lds = (size_t)2 * (3 * p.OW + 10) * 64 * sizeof(float);
        if (nstrip == 5) hipLaunchKernelGGL(groupdw_nhwc_stream_kernel<5>, dim3((unsigned)nb), dim3(320), lds, (hipStream_t)stream, p);
        else             hipLaunchKernelGGL(groupdw_nhwc_stream_kernel<6>, dim3((unsigned)nb), dim3(384), lds, (hipStream_t)stream, p);
        USOT_CHECK_LAUNCH();
        return USOT_OK;
    }
    if (mode == 2) {
        if (p.OH > 32) return USOT_EINVAL;
        p.total = total;
        p.nty = p.ntx = 1;
        const long ncolg = (p.OW + 3) / 4;
        const long nb = p.C == 256 ? 8 * (((total + 1) / 2) * ncolg) : (long)(p.C / 64) * total * ncolg;
        hipLaunchKernelGGL(groupdw_nhwc_col_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
        USOT_CHECK_LAUNCH();
        return USOT_OK;
    }
    p.total = total;
    p.nty = (p.OH + 4) / 5;
    p.ntx = mode == 1 ? p.OW : (p.OW + 4) / 5;   // every other mode: 5x5 patches
    const long ntile = (long)p.nty * p.ntx;
    const long blocks = p.C == 256 ? 8 * ((((total + 1) / 2) * ntile + 3) / 4)
                                   : (long)(p.C / 64) * ((total * ntile + 3) / 4);
    if (blocks > 0x7fffffffL) return USOT_EINVAL;
    // 5 -> 5x5 patches capped at 128 VGPRs (4 waves/SIMD; spills); 50 -> uncapped; 52 -> 2 waves
    if (mode == 1)       hipLaunchKernelGGL((groupdw_nhwc_kernel<5, 1, 2>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (mode == 50) hipLaunchKernelGGL((groupdw_nhwc_kernel<5, 5, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (mode == 52) hipLaunchKernelGGL((groupdw_nhwc_kernel<5, 5, 2>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else                 hipLaunchKernelGGL((groupdw_nhwc_kernel<5, 5, 4>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_groupdw_f32(void *stream, const usot_groupdw_desc *d)
{
    return usot_groupdw_multi_f32(stream, d, 1);
}

extern "C" int usot_groupdw_multi_f32(void *stream, const usot_groupdw_desc *d, int nseg)
{
    return groupdw_multi_impl(stream, d, nseg, 0);
}

/* the same launch with the OUTPUT maps stored as fp16 (out_dtype 1) or bf16 (2): d[i].out points to 16-bit elements.  fp32
 * arithmetic, one rounding at the store.  25- and 27-wide responses (the LDS-DMA kernel). */
extern "C" int usot_groupdw_multi_lp(void *stream, const usot_groupdw_desc *d, int nseg, int out_dtype)
{
    if (out_dtype != 1 && out_dtype != 2) return USOT_EINVAL;
    return groupdw_multi_impl(stream, d, nseg, out_dtype);
}
