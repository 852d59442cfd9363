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
struct GdwK {
    const float *x[3];
    const float *z[3];
    float *out;
    int x_cs[3], x_co[3], z_cs[3], z_co[3];
    float wsm[3];
    int S, x_rep, OH, OW, C;
    int ncg;      // column groups
};

// branch geometry is fixed by the 7x7 template and the three encoder dilations
template <int B> struct Geo;
template <> struct Geo<0> { static constexpr int HK = 5, WK = 5; };
template <> struct Geo<1> { static constexpr int HK = 3, WK = 5; };
template <> struct Geo<2> { static constexpr int HK = 5, WK = 3; };

template <int CW>
__global__ __launch_bounds__(256) void groupdw_nhwc_kernel(const GdwK p)
{
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int cgp = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int s = blockIdx.z;
    if (cgp >= p.ncg || c >= p.C) return;
    const int j0 = cgp * CW;
    const int xs = s / p.x_rep;

    // taps, pre-scaled by softmax(weight)[b]
    float k0[5][5], k1[3][5], k2[5][3];
    {
        const float *z0 = p.z[0] + (long)s * 25 * p.z_cs[0] + p.z_co[0] + c;
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int v = 0; v < 5; ++v) k0[u][v] = p.wsm[0] * z0[(u * 5 + v) * p.z_cs[0]];
        const float *z1 = p.z[1] + (long)s * 15 * p.z_cs[1] + p.z_co[1] + c;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 5; ++v) k1[u][v] = p.wsm[1] * z1[(u * 5 + v) * p.z_cs[1]];
        const float *z2 = p.z[2] + (long)s * 15 * p.z_cs[2] + p.z_co[2] + c;
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) k2[u][v] = p.wsm[2] * z2[(u * 3 + v) * p.z_cs[2]];
    }
    const int W0 = p.OW + 4, W1 = p.OW + 4, W2 = p.OW + 2;      // search-map widths
    const int H0 = p.OH + 4, H1 = p.OH + 2, H2 = p.OH + 4;
    const float *x0 = p.x[0] + (long)xs * H0 * W0 * p.x_cs[0] + p.x_co[0] + c;
    const float *x1 = p.x[1] + (long)xs * H1 * W1 * p.x_cs[1] + p.x_co[1] + c;
    const float *x2 = p.x[2] + (long)xs * H2 * W2 * p.x_cs[2] + p.x_co[2] + c;
    float *o = p.out + ((long)s * p.OH * p.OW) * p.C + c;

    // A[u][j]: partial sum of output row (r - u) after consuming input row r
    float A[5][CW];
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int j = 0; j < CW; ++j) A[u][j] = 0.f;

    for (int r = 0; r < p.OH + 4; ++r) {
        {   // 5x5 branch: row r of x0
            float xv[CW + 4];
#pragma unroll
            for (int q = 0; q < CW + 4; ++q) {
                const int col = j0 + q;
                xv[q] = col < W0 ? x0[((long)r * W0 + col) * p.x_cs[0]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int j = 0; j < CW; ++j)
#pragma unroll
                    for (int v = 0; v < 5; ++v) A[u][j] = fmaf(xv[j + v], k0[u][v], A[u][j]);
        }
        if (r < H1) {   // 3x5 branch
            float xv[CW + 4];
#pragma unroll
            for (int q = 0; q < CW + 4; ++q) {
                const int col = j0 + q;
                xv[q] = col < W1 ? x1[((long)r * W1 + col) * p.x_cs[1]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int j = 0; j < CW; ++j)
#pragma unroll
                    for (int v = 0; v < 5; ++v) A[u][j] = fmaf(xv[j + v], k1[u][v], A[u][j]);
        }
        {   // 5x3 branch
            float xv[CW + 2];
#pragma unroll
            for (int q = 0; q < CW + 2; ++q) {
                const int col = j0 + q;
                xv[q] = col < W2 ? x2[((long)r * W2 + col) * p.x_cs[2]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int j = 0; j < CW; ++j)
#pragma unroll
                    for (int v = 0; v < 3; ++v) A[u][j] = fmaf(xv[j + v], k2[u][v], A[u][j]);
        }
        const int done = r - 4;       // output row finished by this input row
        if (done >= 0) {
#pragma unroll
            for (int j = 0; j < CW; ++j)
                if (j0 + j < p.OW) o[((long)done * p.OW + j0 + j) * p.C] = A[4][j];
        }
#pragma unroll
        for (int u = 4; u > 0; --u)
#pragma unroll
            for (int j = 0; j < CW; ++j) A[u][j] = A[u - 1][j];
#pragma unroll
        for (int j = 0; j < CW; ++j) A[0][j] = 0.f;
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

extern "C" int usot_groupdw_f32(void *stream, const usot_groupdw_desc *d)
{
    if (!d || !d->out || d->S <= 0 || d->x_rep < 1 || d->OH < 1 || d->OW < 1) return USOT_EINVAL;
    if (d->C <= 0 || (d->C & 63)) return USOT_EINVAL;
    static const int hk[3] = {5, 3, 5}, wk[3] = {5, 5, 3};
    GdwK p;
    for (int b = 0; b < 3; ++b) {
        if (!d->x[b] || !d->z[b] || d->hk[b] != hk[b] || d->wk[b] != wk[b]) return USOT_EINVAL;
        p.x[b] = d->x[b]; p.z[b] = d->z[b];
        p.x_cs[b] = d->x_cs[b] > 0 ? d->x_cs[b] : d->C; p.x_co[b] = d->x_co[b];
        p.z_cs[b] = d->z_cs[b] > 0 ? d->z_cs[b] : d->C; p.z_co[b] = d->z_co[b];
        p.wsm[b] = d->wsm[b];
    }
    p.out = d->out; p.S = d->S; p.x_rep = d->x_rep; p.OH = d->OH; p.OW = d->OW; p.C = d->C;
    int cw = d->cols_per_thread;
    if (cw == 0) cw = (d->S >= 64 && d->OW % 5 == 0) ? 5 : 1;
    if (cw != 1 && cw != 5) return USOT_EINVAL;
    p.ncg = (d->OW + cw - 1) / cw;
    dim3 grid((p.ncg + 3) / 4, d->C / 64, d->S);
    hipStream_t s = (hipStream_t)stream;
    if (cw == 5) hipLaunchKernelGGL(groupdw_nhwc_kernel<5>, grid, dim3(256), 0, s, p);
    else         hipLaunchKernelGGL(groupdw_nhwc_kernel<1>, grid, dim3(256), 0, s, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}
