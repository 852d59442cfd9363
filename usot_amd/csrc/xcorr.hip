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

template <int RS, int CW>
__global__ __launch_bounds__(256) void groupdw_nhwc_kernel(const GdwK p)
{
    // Work unit = (sample, patch); channel group = 64 lanes.  With C = 256 the 1-D grid is
    // laid out so that XCD x (= block id % 8, observed dispatch order; speed only) always
    // works on channel group x % 4: each XCD's L2 then holds one quarter of the search maps
    // instead of every XCD pulling all of them across the fabric.
    const int ntile = p.nty * p.ntx;
    const int units = p.total * ntile;
    const int wave = threadIdx.x >> 6;
    int cgi, unit;
    if (p.C == 256) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        cgi = xcd & 3;
        unit = (slot * 4 + wave) * 2 + (xcd >> 2);
    } else {
        const int ncg = p.C >> 6;
        cgi = blockIdx.x % ncg;
        unit = (blockIdx.x / ncg) * 4 + wave;
    }
    if (unit >= units) return;
    const int c = cgi * 64 + (threadIdx.x & 63);
    int s = unit / ntile, sg = 0;
    const int tile = unit - s * ntile;
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

extern "C" int usot_groupdw_multi_f32(void *stream, const usot_groupdw_desc *d, int nseg)
{
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
    const int mode = d[0].cols_per_thread;      // 0/5: 5x5 patches; 1: 5x1 strips (more waves)
    hipStream_t s = (hipStream_t)stream;
    if (mode != 0 && mode != 1 && mode != 5) return USOT_EINVAL;
    p.total = total;
    p.nty = (p.OH + 4) / 5;
    p.ntx = mode == 1 ? p.OW : (p.OW + 4) / 5;
    const long units = (long)total * p.nty * p.ntx;
    const long blocks = p.C == 256 ? 8 * (((units + 1) / 2 + 3) / 4) : (long)(p.C / 64) * ((units + 3) / 4);
    if (blocks > 0x7fffffffL) return USOT_EINVAL;
    if (mode == 1) hipLaunchKernelGGL((groupdw_nhwc_kernel<5, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else           hipLaunchKernelGGL((groupdw_nhwc_kernel<5, 5>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_groupdw_f32(void *stream, const usot_groupdw_desc *d)
{
    return usot_groupdw_multi_f32(stream, d, 1);
}
