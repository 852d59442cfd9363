// Small head-side kernels: Conf_Fusion reduction, Precise RoI Pooling forward, layout
// permutes at the API edge, and the on-device decode of a frame's response maps.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include "usot_hip.h"
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- Conf_Fusion (connect.py:132-142): out = sum_m conf_m * value_m / sum_m conf_m -------
// OT: 0 = fp32 output, 1 = fp16, 2 = bf16 (`out` then points to 16-bit elements); IT likewise for the INPUT map cv
template <int IT> __device__ __forceinline__ f32x4 cf_load4(const void *base, long e)
{
    if constexpr (IT == 0) {
        return *(const f32x4 *)((const float *)base + e);
    } else {
        const uint2 v = *(const uint2 *)((const uint16_t *)base + e);
        f32x4 r;
        if constexpr (IT == 1) {
            r[0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v.x & 0xffffu)); r[1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v.x >> 16));
            r[2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v.y & 0xffffu)); r[3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v.y >> 16));
        } else {
            r[0] = __builtin_bit_cast(float, v.x << 16); r[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
            r[2] = __builtin_bit_cast(float, v.y << 16); r[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
        }
        return r;
    }
}

template <int OT, int IT = 0>
__global__ __launch_bounds__(256) void conf_fusion_reduce_kernel(
    const void *__restrict__ cv, float *__restrict__ out, int B, int M, int P, int C4)
{
    const long total = (long)B * P * C4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % C4);
        const long bp = idx / C4;
        const int pix = (int)(bp % P);
        const int b = (int)(bp / P);
        f32x4 den = {0.f, 0.f, 0.f, 0.f}, num = {0.f, 0.f, 0.f, 0.f};
        const long base = ((long)b * M * P + pix) * 2 * C4 * 4 + c * 4;        // element index of this lane's four conf values
        const long mstride = (long)P * 2 * C4 * 4;
        for (int m = 0; m < M; ++m) den += cf_load4<IT>(cv, base + m * mstride);
        // same association as the reference: normalise each conf, then weight and add
        for (int m = 0; m < M; ++m) {
            const f32x4 c4 = cf_load4<IT>(cv, base + m * mstride);        // L1/L2 hit
            const f32x4 v4 = cf_load4<IT>(cv, base + m * mstride + C4 * 4);
            num += (c4 / den) * v4;
        }
        if constexpr (OT == 0) {
            *(f32x4 *)(out + idx * 4) = num;
        } else {
            uint16_t h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                h[e] = OT == 1 ? __builtin_bit_cast(uint16_t, (_Float16)num[e]) : __builtin_bit_cast(uint16_t, (__bf16)num[e]);
            uint2 o2;
            o2.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
            o2.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
            *(uint2 *)((uint16_t *)out + idx * 4) = o2;
        }
    }
}

// ---- Precise RoI Pooling forward (prroi_pooling_gpu_impl.cu:37-42,71-106,149-212) --------
struct PrK {
    const float *feat, *rois;
    float *out;
    int R, C, H, W, PH, PW;
    float scale;
    long f_sb, f_sc, f_sh, f_sw, o_sr, o_sc, o_sh, o_sw;
};

__device__ __forceinline__ float pr_tap(const float *d, int h, int w, const PrK &p)
{
    if (h < 0 || w < 0 || h >= p.H || w >= p.W) return 0.f;
    return d[h * p.f_sh + w * p.f_sw];
}

// Closed-form integral of the bilinear hat over one unit cell (prroi_pooling_gpu_impl.cu:71-106).
// With t measured from a cell corner, the 1-D weight of that corner over [a, b] is
// W(a, b) = int_a^b (1 - t) dt = (b - b*b/2) - (a - a*a/2).  A cell needs only two W per axis (near
// and far corner); the four corner terms are their outer product, accumulated in the order
// (h0,w0), (h0,w1), (h1,w0), (h1,w1) — the per-term float32 rounding of the reference's launcher.
__device__ __forceinline__ float pr_hat(float a, float b)
{
    return b - 0.5f * b * b - a + 0.5f * a * a;
}

__device__ __forceinline__ float pr_cell(const float *d, int h0, int w0, float y0, float x0,
                                         float y1, float x1, const PrK &p)
{
    const int h1 = h0 + 1, w1 = w0 + 1;
    const float wx_near = pr_hat(x0 - (float)w0, x1 - (float)w0);
    const float wx_far  = pr_hat((float)w1 - x1, (float)w1 - x0);
    const float wy_near = pr_hat(y0 - (float)h0, y1 - (float)h0);
    const float wy_far  = pr_hat((float)h1 - y1, (float)h1 - y0);
    float sum = pr_tap(d, h0, w0, p) * (wx_near * wy_near);
    sum += pr_tap(d, h0, w1, p) * (wx_far * wy_near);
    sum += pr_tap(d, h1, w0, p) * (wx_near * wy_far);
    sum += pr_tap(d, h1, w1, p) * (wx_far * wy_far);
    return sum;
}

// channel is the fastest-varying index of the thread map: coalesced on NHWC features
__global__ __launch_bounds__(256) void prroi_forward_kernel(const PrK p)
{
    const long total = (long)p.R * p.PH * p.PW * p.C;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % p.C);
        long r = idx / p.C;
        const int pw = (int)(r % p.PW); r /= p.PW;
        const int ph = (int)(r % p.PH);
        const int n = (int)(r / p.PH);
        const float *roi = p.rois + (long)n * 5;
        const int b = (int)roi[0];
        const float rsw = roi[1] * p.scale, rsh = roi[2] * p.scale;
        const float rew = roi[3] * p.scale, reh = roi[4] * p.scale;
        const float rw = fmaxf(rew - rsw, 0.f), rh = fmaxf(reh - rsh, 0.f);
        const float bh = rh / (float)p.PH, bw = rw / (float)p.PW;
        const float win = fmaxf(0.f, bw * bh);
        float *o = p.out + n * p.o_sr + c * p.o_sc + ph * p.o_sh + pw * p.o_sw;
        if (win == 0.f) { *o = 0.f; continue; }
        const float *d = p.feat + b * p.f_sb + c * p.f_sc;
        const float wsw = rsw + bw * pw, wsh = rsh + bh * ph;
        const float wew = wsw + bw, weh = wsh + bh;
        const int s_w = (int)floorf(wsw), e_w = (int)ceilf(wew);
        const int s_h = (int)floorf(wsh), e_h = (int)ceilf(weh);
        float sum = 0.f;
        for (int wi = s_w; wi < e_w; ++wi)
            for (int hi = s_h; hi < e_h; ++hi)
                sum += pr_cell(d, hi, wi, fmaxf(wsh, (float)hi), fmaxf(wsw, (float)wi),
                               fminf(weh, (float)hi + 1.0f), fminf(wew, (float)wi + 1.0f), p);
        *o = sum / win;
    }
}

// ---- Precise RoI Pooling gradients (training side of the operator; prroi_pooling_gpu_impl.cu:214-380) ---------
// out = (1/A) * hy^T W hx with hx[i] = int_bin hat(x - i) dx: separable.  pr_axis_weight() is that integral for one
// pixel in float32, assembled from the same two corner pieces (pr_hat) the forward uses per unit cell: the piece of
// the cell on the pixel's left, where the pixel is the far corner, and of the cell on its right, where it is the
// near one.  Contiguous NCHW as the reference's ABI.
struct PrGradK {
    const float *feat, *rois, *top, *gout;
    float *gin;                              // feature gradient [B][C][H][W]  or  RoI gradient [R][5]
    int R, B, C, H, W, PH, PW;
    float scale;
};

__device__ __forceinline__ float pr_axis_weight(int i, float lo, float hi, int s, int e)
{
    float w = 0.f;
    if (i > s) w += pr_hat((float)i - fminf(hi, (float)i), (float)i - fmaxf(lo, (float)(i - 1)));       // cell i-1, far corner
    if (i < e) w += pr_hat(fmaxf(lo, (float)i) - (float)i, fminf(hi, (float)(i + 1)) - (float)i);         // cell i, near corner
    return w;
}

struct PrBin { int b; float x0, y0, x1, y1, area; int s_w, e_w, s_h, e_h; };

__device__ __forceinline__ PrBin pr_bin(const float *roi, float scale, int PH, int PW, int ph, int pw)
{
    PrBin q;
    q.b = (int)roi[0];
    const float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
    const float bh = fmaxf(reh - rsh, 0.f) / (float)PH, bw = fmaxf(rew - rsw, 0.f) / (float)PW;
    q.area = fmaxf(0.f, bw * bh);
    q.x0 = rsw + bw * pw; q.y0 = rsh + bh * ph;
    q.x1 = q.x0 + bw;     q.y1 = q.y0 + bh;
    q.s_w = (int)floorf(q.x0); q.e_w = (int)ceilf(q.x1);
    q.s_h = (int)floorf(q.y0); q.e_h = (int)ceilf(q.y1);
    return q;
}

// Feature gradient: one thread per pooled element (pw fastest: coalesced top_diff), ONE atomic per touched pixel
// (the reference issues one per cell corner, up to four per pixel): gin[b][c][y][x] += g / A * hy[y] * hx[x].
__global__ __launch_bounds__(256) void prroi_backward_kernel(const PrGradK p)
{
    const long total = (long)p.R * p.C * p.PH * p.PW;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int pw = (int)(idx % p.PW);
        long r = idx / p.PW;
        const int ph = (int)(r % p.PH); r /= p.PH;
        const int c = (int)(r % p.C);
        const int n = (int)(r / p.C);
        const PrBin q = pr_bin(p.rois + (long)n * 5, p.scale, p.PH, p.PW, ph, pw);
        if (q.area == 0.f || q.b < 0 || q.b >= p.B) continue;
        const float share = p.gout[idx] / q.area;
        if (share == 0.f) continue;
        float *d = p.gin + ((long)q.b * p.C + c) * p.H * p.W;
        const int y_lo = q.s_h < 0 ? 0 : q.s_h, y_hi = q.e_h >= p.H ? p.H - 1 : q.e_h;
        const int x_lo = q.s_w < 0 ? 0 : q.s_w, x_hi = q.e_w >= p.W ? p.W - 1 : q.e_w;
        for (int y = y_lo; y <= y_hi; ++y) {
            const float wy = share * pr_axis_weight(y, q.y0, q.y1, q.s_h, q.e_h);
            if (wy == 0.f) continue;
            for (int x = x_lo; x <= x_hi; ++x) {
                const float v = wy * pr_axis_weight(x, q.x0, q.x1, q.s_w, q.e_w);
                if (v != 0.f) atomicAdd(d + y * p.W + x, v);
            }
        }
    }
}

// zero-extended bilinear surface along one axis: value at (row y, real x) and (real y, column x)
__device__ __forceinline__ float pr_lerp_x(const float *d, int y, float x, int H, int W)
{
    if (y < 0 || y >= H) return 0.f;
    const int x0 = (int)floorf(x);
    const float a = x - (float)x0;
    const float v0 = (x0 >= 0 && x0 < W) ? d[y * W + x0] : 0.f, v1 = (x0 + 1 >= 0 && x0 + 1 < W) ? d[y * W + x0 + 1] : 0.f;
    return (1.f - a) * v0 + a * v1;
}
__device__ __forceinline__ float pr_lerp_y(const float *d, float y, int x, int H, int W)
{
    if (x < 0 || x >= W) return 0.f;
    const int y0 = (int)floorf(y);
    const float a = y - (float)y0;
    const float v0 = (y0 >= 0 && y0 < H) ? d[y0 * W + x] : 0.f, v1 = (y0 + 1 >= 0 && y0 + 1 < H) ? d[(y0 + 1) * W + x] : 0.f;
    return (1.f - a) * v0 + a * v1;
}

// RoI gradient.  Leibniz: moving a bin edge changes the integral by the line integral of the surface along that edge,
// L_x(x) = sum_y hy[y] * f(x, y); d out / d xs = (-L_x(xs) + (ye - ys) out) / A, d out / d xe = (L_x(xe) - (ye - ys) out) / A
// (same in y), and the edges are affine in the RoI corners.  blockIdx.y = RoI, so a wavefront reduces its four partial
// sums with shuffles and issues four atomics (the reference: four per pooled element onto the same four words).
__global__ __launch_bounds__(256) void prroi_coor_backward_kernel(const PrGradK p)
{
    const int n = blockIdx.y;
    const int per_roi = p.C * p.PH * p.PW;
    float g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < per_roi; e += gridDim.x * 256) {
        const int pw = e % p.PW, ph = (e / p.PW) % p.PH, c = e / (p.PW * p.PH);
        const PrBin q = pr_bin(p.rois + (long)n * 5, p.scale, p.PH, p.PW, ph, pw);
        if (q.area == 0.f || q.b < 0 || q.b >= p.B) continue;
        const long idx = (long)n * per_roi + e;
        const float g = p.gout[idx];
        if (g == 0.f) continue;                                       // .cu:315-317
        const float top = p.top[idx];
        const float *d = p.feat + ((long)q.b * p.C + c) * p.H * p.W;
        float lx0 = 0.f, lx1 = 0.f, ly0 = 0.f, ly1 = 0.f;
        for (int y = q.s_h; y <= q.e_h; ++y) {
            const float wy = pr_axis_weight(y, q.y0, q.y1, q.s_h, q.e_h);
            lx0 += wy * pr_lerp_x(d, y, q.x0, p.H, p.W);
            lx1 += wy * pr_lerp_x(d, y, q.x1, p.H, p.W);
        }
        for (int x = q.s_w; x <= q.e_w; ++x) {
            const float wx = pr_axis_weight(x, q.x0, q.x1, q.s_w, q.e_w);
            ly0 += wx * pr_lerp_y(d, q.y0, x, p.H, p.W);
            ly1 += wx * pr_lerp_y(d, q.y1, x, p.H, p.W);
        }
        const float k = p.scale / q.area * g;
        const float dxs = (-lx0 + (q.y1 - q.y0) * top) * k, dxe = (lx1 - (q.y1 - q.y0) * top) * k;
        const float dys = (-ly0 + (q.x1 - q.x0) * top) * k, dye = (ly1 - (q.x1 - q.x0) * top) * k;
        const float fx0 = (float)pw / (float)p.PW, fx1 = (float)(pw + 1) / (float)p.PW;
        const float fy0 = (float)ph / (float)p.PH, fy1 = (float)(ph + 1) / (float)p.PH;
        g1 += dxs * (1.f - fx0) + dxe * (1.f - fx1);
        g2 += dys * (1.f - fy0) + dye * (1.f - fy1);
        g3 += dxs * fx0 + dxe * fx1;
        g4 += dys * fy0 + dye * fy1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        g1 += __shfl_xor(g1, o); g2 += __shfl_xor(g2, o); g3 += __shfl_xor(g3, o); g4 += __shfl_xor(g4, o);
    }
    if ((threadIdx.x & 63) == 0) {
        float *gr = p.gin + (long)n * 5;
        if (g1 != 0.f) atomicAdd(gr + 1, g1);
        if (g2 != 0.f) atomicAdd(gr + 2, g2);
        if (g3 != 0.f) atomicAdd(gr + 3, g3);
        if (g4 != 0.f) atomicAdd(gr + 4, g4);
    }
}

// ---- generic 4-D permute: dst dense [D0][D1][D2][D3] <- strided src ------------------------
__global__ __launch_bounds__(256) void permute4_kernel(
    const float *__restrict__ src, float *__restrict__ dst, int D1, int D2, int D3, long total,
    long s0, long s1, long s2, long s3)
{
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % D3);
        long r = idx / D3;
        const int b = (int)(r % D2); r /= D2;
        const int a = (int)(r % D1);
        const long n = r / D1;
        dst[idx] = src[n * s0 + a * s1 + b * s2 + c * s3];
    }
}

// ---- row gather / scatter with DEVICE-resident indices: lets a captured graph pick the
// memory-queue kernels of a frame (usot_tracker.py:222-256) and append the new one --------
__global__ __launch_bounds__(256) void rows_copy_kernel(
    const float *__restrict__ src, const int *__restrict__ idx, float *__restrict__ dst,
    int n_rows, int row_len4, int scatter)
{
    const long total = (long)n_rows * row_len4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / row_len4), e = (int)(i - (long)r * row_len4);
        const long sr = scatter ? r : idx[r], dr = scatter ? idx[r] : r;
        ((f32x4 *)dst)[dr * row_len4 + e] = ((const f32x4 *)src)[sr * row_len4 + e];
    }
}

// ---- thin 3x3 convolutions: the prediction heads (connect.py:236-241,275): bbox_pred 256 -> 4,
// cls_pred / cls_memory_pred 256 -> 1.  On the tile kernels these are 2304-deep dot products with
// 1-4 useful output columns of a 32-wide tile (21 us per frame, 0.9 TFLOP/s).  Here one wavefront
// owns one output pixel of one problem: lane = 4 input channels, 9 taps x 16-byte loads of x and of
// each filter row, a butterfly reduction, lane 0 stores.  Up to four problems (x groups) per launch.
struct ThinK {
    const float *x[4], *w[4], *bias[4];
    float *y[4];
    int cout[4], act[4], groups[4], start[5];      // start: first work item of problem i
    int ycs[4], yco[4];                            // channels of the output tensor, first channel this problem writes (a 4-channel problem may run as two of 2)
    long x_gs[4], w_gs[4], y_gs[4];
    int b_gs[4];
    int n, N, H, W, Cin;                           // shared geometry (3x3, stride 1, pad 1)
};

__device__ __forceinline__ float thin_act(float v, int a)
{
    switch (a) {
    case USOT_ACT_RELU: return fmaxf(v, 0.0f);
    case USOT_ACT_EXP:  return expf(v);
    case USOT_ACT_CONF: return expf(fminf(fmaxf(v, 0.0f), 4.0f));
    default:            return v;
    }
}

template <int CO>
__device__ __forceinline__ void thin_pixel(const ThinK &k, int pi, int g, int pix, int lane)
{
    const int P = k.H * k.W;
    const int n = pix / P, r = pix - n * P;
    const int oh = r / k.W, ow = r - oh * k.W;
    const float *xg = k.x[pi] + (long)g * k.x_gs[pi] + (long)n * P * k.Cin;
    const float *wg = k.w[pi] + (long)g * k.w_gs[pi];
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    for (int c0 = lane * 4; c0 < k.Cin; c0 += 256) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // branch-free: padding taps read a clamped (valid) pixel and are scaled by 0, so that all 9
            // pixel loads and 9 x CO filter loads of the lane can be in flight together
            const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
            const bool ok = (unsigned)ih < (unsigned)k.H && (unsigned)iw < (unsigned)k.W;
            const int ihc = min(max(ih, 0), k.H - 1), iwc = min(max(iw, 0), k.W - 1);
            f32x4 xv = *(const f32x4 *)(xg + ((long)ihc * k.W + iwc) * k.Cin + c0);
            xv *= ok ? 1.f : 0.f;
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                const f32x4 wv = *(const f32x4 *)(wg + ((long)c * 9 + t) * k.Cin + c0);
                acc[c] = fmaf(xv[0], wv[0], fmaf(xv[1], wv[1], fmaf(xv[2], wv[2], fmaf(xv[3], wv[3], acc[c]))));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
    }
    if (lane == 0) {
        const float *bg = k.bias[pi] ? k.bias[pi] + (long)g * k.b_gs[pi] : nullptr;
        float *yg = k.y[pi] + (long)g * k.y_gs[pi];
#pragma unroll
        for (int c = 0; c < CO; ++c)
            yg[((long)n * k.ycs[pi] + k.yco[pi] + c) * P + r] = thin_act(acc[c] + (bg ? bg[c] : 0.f), k.act[pi]);   // NCHW
    }
}

__global__ __launch_bounds__(256) void thin_conv3x3_kernel(const ThinK k)
{
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= k.start[k.n]) return;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < k.n && item >= k.start[q]) pi = q;
    const int local = item - k.start[pi];
    const int per_g = k.N * k.H * k.W;
    const int g = local / per_g, pix = local - g * per_g;
    switch (k.cout[pi]) {
    case 1: thin_pixel<1>(k, pi, g, pix, lane); break;
    case 2: thin_pixel<2>(k, pi, g, pix, lane); break;
    case 3: thin_pixel<3>(k, pi, g, pix, lane); break;
    default: thin_pixel<4>(k, pi, g, pix, lane); break;
    }
}

// The same arithmetic for BATCHES (round 4): one wavefront per output pixel re-reads 9 x 1 KB of activations and 9 x CO KB of
// filters per pixel — at 32 streams (20 000 pixels x 3 problems) that is 0.5 GB + 1 GB of L1 / L2 traffic and 180 us.  Here a
// wavefront owns a ROW of output pixels: the filters live in registers (lane = 4 input channels: 9 x CO float4), a 3 x 3
// window of float4 slides along the row (three new loads per pixel instead of nine), the next column is loaded before the
// current pixel is accumulated.  Per lane the same taps in the same order with the same fused multiply-adds, then the same
// butterfly: bit-identical to thin_pixel (out-of-image taps contribute +0 either way).  Cin = 256 (one float4 per lane).
template <int CO>
__device__ __forceinline__ void thin_row(const ThinK &k, int pi, int g, int nrow, int lane)
{
    const int P = k.H * k.W;
    const int n = nrow / k.H, oh = nrow - n * k.H;
    const float *xg = k.x[pi] + (long)g * k.x_gs[pi] + (long)n * P * k.Cin + lane * 4;
    const float *wg = k.w[pi] + (long)g * k.w_gs[pi] + lane * 4;
    f32x4 wv[CO][9];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[c][t] = *(const f32x4 *)(wg + ((long)c * 9 + t) * k.Cin);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    auto column = [&](int iw, f32x4 (&col)[3]) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ih = oh + dy - 1;
            const bool ok = (unsigned)ih < (unsigned)k.H && (unsigned)iw < (unsigned)k.W;
            col[dy] = ok ? *(const f32x4 *)(xg + ((long)ih * k.W + iw) * k.Cin) : zero;
        }
    };
    f32x4 c0[3], c1[3], c2[3], c3[3];
    c0[0] = c0[1] = c0[2] = zero;                   // column -1
    column(0, c1);
    column(1, c2);
    const float *bg = k.bias[pi] ? k.bias[pi] + (long)g * k.b_gs[pi] : nullptr;
    float *yg = k.y[pi] + (long)g * k.y_gs[pi];
    float keep[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) keep[c] = 0.f;
    for (int ow = 0; ow < k.W; ++ow) {
        column(ow + 2, c3);                          // in flight under this pixel's arithmetic
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4 xv = (t % 3 == 0) ? c0[t / 3] : ((t % 3 == 1) ? c1[t / 3] : c2[t / 3]);
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                const f32x4 w4 = wv[c][t];
                acc[c] = fmaf(xv[0], w4[0], fmaf(xv[1], w4[1], fmaf(xv[2], w4[2], fmaf(xv[3], w4[3], acc[c]))));
            }
        }
#pragma unroll
        for (int c = 0; c < CO; ++c) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
        }
        // the xor butterfly leaves the sum in every lane: lane `ow` keeps pixel ow's, the row is stored once per channel below
        // (25 x CO four-byte stores from lane 0 were a write transaction each)
#pragma unroll
        for (int c = 0; c < CO; ++c) keep[c] = lane == ow ? acc[c] : keep[c];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) { c0[dy] = c1[dy]; c1[dy] = c2[dy]; c2[dy] = c3[dy]; }
    }
    if (lane < k.W) {
#pragma unroll
        for (int c = 0; c < CO; ++c)
            yg[((long)n * k.ycs[pi] + k.yco[pi] + c) * P + oh * k.W + lane] = thin_act(keep[c] + (bg ? bg[c] : 0.f), k.act[pi]);   // NCHW
    }
}

// work item = (problem, group, image, output row); k.start counts rows here.  MAXCO = the widest problem of the launch: the
// kernel's register allocation is that of its widest case (CO = 4: 243 registers, two waves per SIMD for EVERY problem of the
// launch), so the launcher splits 4-channel problems in two and runs the <2> instantiation (~half the registers)
template <int MAXCO>
__global__ __launch_bounds__(256) void thin_conv3x3_rows_kernel(const ThinK k)
{
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= k.start[k.n]) return;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < k.n && item >= k.start[q]) pi = q;
    const int local = item - k.start[pi];
    const int per_g = k.N * k.H;
    const int g = local / per_g, nrow = local - g * per_g;
    if constexpr (MAXCO <= 2) {
        if (k.cout[pi] == 1) thin_row<1>(k, pi, g, nrow, lane);
        else                 thin_row<2>(k, pi, g, nrow, lane);
    } else {
        switch (k.cout[pi]) {
        case 1: thin_row<1>(k, pi, g, nrow, lane); break;
        case 2: thin_row<2>(k, pi, g, nrow, lane); break;
        case 3: thin_row<3>(k, pi, g, nrow, lane); break;
        default: thin_row<4>(k, pi, g, nrow, lane); break;
        }
    }
}

// up to four banks with different row lengths, same row indices, one launch (blockIdx.y = bank)
struct RowsK {
    const float *src[4];
    float *dst[4];
    int row_len4[4];
    int nseg, n_rows, scatter;
    int *stash;        // gather only: idx[n_rows .. n_rows+2] are copied here (see usot_hip.h)
};
__global__ __launch_bounds__(256) void rows_copy_multi_kernel(const RowsK k, const int *__restrict__ idx)
{
    const int sgm = blockIdx.y;
    if (k.stash && blockIdx.x == 0 && sgm == 0 && threadIdx.x < 3) k.stash[threadIdx.x] = idx[k.n_rows + threadIdx.x];
    const int rl = k.row_len4[sgm];
    const f32x4 *__restrict__ src = (const f32x4 *)k.src[sgm];
    f32x4 *__restrict__ dst = (f32x4 *)k.dst[sgm];
    const long total = (long)k.n_rows * rl;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / rl), e = (int)(i - (long)r * rl);
        const long sr = k.scatter ? r : idx[r], dr = k.scatter ? idx[r] : r;
        dst[dr * rl + e] = src[sr * rl + e];
    }
}

// Append + gather in ONE launch (round 6, the session's 'defer_append' = 2 frame): the row the PREVIOUS frame pooled (bank 0)
// and its three encodings (banks 1-3) go to row idx[slot_pos] of their banks, and the n_pick rows idx[0 .. n_pick) of banks
// 1-3 are gathered into the frame's picked-kernel buffers - a picked row that IS the appended row is read from the new
// row itself, so the two halves need no ordering between them.  blockIdx.y: 0-3 = append bank y, 4-6 = gather bank y - 3.
// idx lives in pinned HOST memory (the control block): a workgroup fetches its words once.
struct RowsAG {
    const float *fresh[4];
    float *bank[4];
    float *picked[3];
    int row_len4[4];
    int n_pick, slot_pos;
};
__global__ __launch_bounds__(256) void rows_append_gather_kernel(const RowsAG k, const int *__restrict__ idx)
{
    __shared__ int rows[33];
    if (threadIdx.x < k.n_pick) rows[threadIdx.x] = idx[threadIdx.x];
    if (threadIdx.x == 32) rows[32] = idx[k.slot_pos];
    __syncthreads();
    const int slot = rows[32];
    const int job = blockIdx.y;
    if (job < 4) {
        const int rl = k.row_len4[job];
        const f32x4 *__restrict__ src = (const f32x4 *)k.fresh[job];
        f32x4 *__restrict__ dst = (f32x4 *)k.bank[job] + (long)slot * rl;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < rl; i += gridDim.x * 256) dst[i] = src[i];
        return;
    }
    const int b = job - 3;
    const int rl = k.row_len4[b];
    const f32x4 *__restrict__ fresh = (const f32x4 *)k.fresh[b];
    const f32x4 *__restrict__ bank = (const f32x4 *)k.bank[b];
    f32x4 *__restrict__ dst = (f32x4 *)k.picked[b - 1];
    const int total = k.n_pick * rl;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / rl, e = i - r * rl;
        const int row = rows[r];
        dst[i] = row == slot ? fresh[e] : bank[(long)row * rl + e];
    }
}

// ---- SiamFC crop on the device (lib/utils/track_utils.py:30-119): window extraction with
// mean-colour padding, OpenCV-style fixed-point bilinear resize (the arithmetic restated in
// usot_amd/hostutils.py::resize_bilinear_u8) and HWC uint8 -> CHW float32, one thread per
// output pixel.  Replaces ~1 ms of numpy per frame with one small kernel.
struct CropK {
    const unsigned char *im;      // [H][W][3]
    float *out;                   // [3][S][S]
    int H, W, S, win;
    int x0, y0;                   // window origin in image coordinates (may be negative)
    int fill[3];
};

__device__ __forceinline__ void crop_axis(int d, int n_src, int n_dst, int &s0, int &s1, int &w0, int &w1)
{
    // OpenCV's own arithmetic (see hostutils._resize_axis): double scale = 1 / (dst / src), the source
    // coordinate rounded to float BEFORE the floor, float fraction, round-half-even coefficients
    const double scale = 1.0 / ((double)n_dst / (double)n_src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.0f; s = 0; }
    if (s >= n_src - 1) { f = 0.0f; s = n_src - 1; }
    w1 = (int)rintf(f * 2048.0f);
    w0 = (int)rintf((1.0f - f) * 2048.0f);
    s0 = s;
    s1 = min(s + 1, n_src - 1);
}

__device__ __forceinline__ int crop_px(const CropK &p, int wx, int wy, int c)
{
    const int ix = p.x0 + wx, iy = p.y0 + wy;
    if ((unsigned)ix >= (unsigned)p.W || (unsigned)iy >= (unsigned)p.H) return p.fill[c];
    return p.im[((long)iy * p.W + ix) * 3 + c];
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const CropK p)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.S * p.S) return;
    const int dy = idx / p.S, dx = idx - dy * p.S;
    if (p.win == p.S) {
#pragma unroll
        for (int c = 0; c < 3; ++c) p.out[((long)c * p.S + dy) * p.S + dx] = (float)crop_px(p, dx, dy, c);
        return;
    }
    if (p.win == 2 * p.S) {          // exact 2x downscale: cv2.resize switches INTER_LINEAR to INTER_AREA
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int v = (crop_px(p, 2 * dx, 2 * dy, c) + crop_px(p, 2 * dx + 1, 2 * dy, c) +
                           crop_px(p, 2 * dx, 2 * dy + 1, c) + crop_px(p, 2 * dx + 1, 2 * dy + 1, c) + 2) >> 2;
            p.out[((long)c * p.S + dy) * p.S + dx] = (float)v;
        }
        return;
    }
    int xa, xb, wxa, wxb, ya, yb, wya, wyb;
    crop_axis(dx, p.win, p.S, xa, xb, wxa, wxb);
    crop_axis(dy, p.win, p.S, ya, yb, wya, wyb);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const long top = (long)crop_px(p, xa, ya, c) * wxa + (long)crop_px(p, xb, ya, c) * wxb;
        const long bot = (long)crop_px(p, xa, yb, c) * wxa + (long)crop_px(p, xb, yb, c) * wxb;
        long v = ((((long)wya * (top >> 4)) >> 16) + (((long)wyb * (bot >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        p.out[((long)c * p.S + dy) * p.S + dx] = (float)v;
    }
}

// ---- decode (usot_tracker.py:138-163): one workgroup, double precision like the numpy
// reference (its grids are float64, so everything after the float32 sigmoid promotes).
// One response cell per thread at S <= 32 (a single pass: the loads of all six maps in flight
// together), the argmax as a wavefront butterfly + one LDS round over the waves, and the WINNING
// thread — which still holds its cell's box, score and penalty in registers — publishes the
// results (round 6; before: 256 threads x 3 cells, an 8-step LDS tree, thread 0 recomputing the
// winner's cell: 10.8 us per frame in the graph).  Same expressions, same order: bit-identical.
struct DecCell {
    double ps, x1, y1, x2, y2, pen;
    float sc;
    int i;
};

// np.argmax semantics (usot_tracker.py:163): first maximum, and a NaN counts as the maximum (the first NaN wins);
// index 0x7fffffff = "no cell" loses to everything
__device__ __forceinline__ bool dec_better(double ov, int oi, double mv, int mi)
{
    const bool on = ov != ov, mn = mv != mv;
    return oi != 0x7fffffff &&
        (mi == 0x7fffffff || (on && !mn) || (on == mn && (on ? oi < mi : (ov > mv || (ov == mv && oi < mi)))));
}

__global__ __launch_bounds__(1024) void decode_kernel(
    const float *__restrict__ cls, const float *__restrict__ cls_mem, const float *__restrict__ bbox,
    const double *__restrict__ window, double *__restrict__ out, int S, int instance_size, int stride,
    float ratio, double penalty_k, double window_influence, double tw, double th,
    const double *__restrict__ tsz_dev, float *__restrict__ roi_out)
{
    // the control block lives in pinned HOST memory: every word the kernel needs from it is fetched here, in one round trip over
    // the link (the ovf address and the frame tag used to be read at the end, one dependent round trip each)
    unsigned long long oa = 0;
    double tag = 0.0;
    if (tsz_dev) {
        tw = tsz_dev[0]; th = tsz_dev[1];
        oa = ((const unsigned long long *)tsz_dev)[3];
        tag = tsz_dev[6];
    }
    __shared__ double wave_v[16];
    __shared__ int wave_i[16];
    __shared__ int win_i;
    const int n = S * S;
    const double tpad = (tw + th) * 0.5;
    const double tsz = sqrt((tw + tpad) * (th + tpad));
    const double tratio = tw / th;
    DecCell best;
    best.ps = -1e300; best.i = 0x7fffffff;
    best.x1 = best.y1 = best.x2 = best.y2 = best.pen = 0.0; best.sc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i / S, c = i - r * S;
        const double gx = (double)((c - S / 2) * stride + instance_size / 2);
        const double gy = (double)((r - S / 2) * stride + instance_size / 2);
        const float s0 = 1.0f / (1.0f + expf(-cls[i]));
        const float s1 = 1.0f / (1.0f + expf(-cls_mem[i]));
        const float sc = ratio * s0 + (1.0f - ratio) * s1;
        const double x1 = gx - (double)bbox[i], y1 = gy - (double)bbox[n + i];
        const double x2 = gx + (double)bbox[2 * n + i], y2 = gy + (double)bbox[3 * n + i];
        const double w = x2 - x1, h = y2 - y1;
        const double pad = (w + h) * 0.5;
        double sr = sqrt((w + pad) * (h + pad)) / tsz;
        sr = fmax(sr, 1.0 / sr);
        double rr = tratio / (w / h);
        rr = fmax(rr, 1.0 / rr);
        const double pen = exp(-(rr * sr - 1.0) * penalty_k);
        const double ps = pen * (double)sc * (1.0 - window_influence) + window[i] * window_influence;
        if (dec_better(ps, i, best.ps, best.i)) {
            best.ps = ps; best.i = i; best.sc = sc; best.pen = pen;
            best.x1 = x1; best.y1 = y1; best.x2 = x2; best.y2 = y2;
        }
    }
    double rv = best.ps;
    int ri = best.i;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(rv, off, 64);
        const int oi = __shfl_xor(ri, off, 64);
        if (dec_better(ov, oi, rv, ri)) { rv = ov; ri = oi; }
    }
    const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { wave_v[wv] = rv; wave_i[wv] = ri; }
    __syncthreads();
    if (threadIdx.x < 64) {
        rv = threadIdx.x < nw ? wave_v[threadIdx.x] : -1e300;
        ri = threadIdx.x < nw ? wave_i[threadIdx.x] : 0x7fffffff;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const double ov = __shfl_xor(rv, off, 64);
            const int oi = __shfl_xor(ri, off, 64);
            if (dec_better(ov, oi, rv, ri)) { rv = ov; ri = oi; }
        }
        if (threadIdx.x == 0) win_i = ri;
    }
    __syncthreads();
    // `win_i` is a valid cell whatever the maps hold (every cell beats "no cell"); its owner publishes
    if (best.i == win_i) {
        out[0] = (double)best.i;
        out[1] = (double)best.sc;
        out[2] = best.pen;
        out[3] = best.x1; out[4] = best.y1; out[5] = best.x2; out[6] = best.y2;
        out[7] = best.ps;
        if (roi_out) {
            // usot_tracker.py:329-350 (pool_label_search): the S-point axis of the response
            // map applied to the search feature; box rounded to float32 first (np.array(...,
            // np.float32)), arithmetic in float64, result stored as float32 (.float()).
            const double lo = (double)((0 - S / 2) * stride + instance_size / 2);
            const double hi = (double)((S - 1 - S / 2) * stride + instance_size / 2);
            const double slope = (double)(2 * (S / 2)) / (hi - lo);
            const double gap = 1.0 / slope;
            const double bx[4] = {best.x1, best.y1, best.x2, best.y2};
            roi_out[0] = 0.f;
            for (int e = 0; e < 4; ++e) {
                double v = (double)(float)bx[e];
                v = fmin(fmax(v, lo - gap), hi + gap);
                roi_out[1 + e] = (float)((v - lo) * slope);
            }
        }
        if (tsz_dev) {
            // tsz_dev[3] (as a 64-bit integer): 0, or the address of the frame's sticky split-fp16 "a sum was not finite" word
            // (usot_conv_desc.ovf).  Its value travels with the results (out[9]) and the word is cleared for the next frame.
            if (oa) {                                 // (then `out` has 10 doubles)
                int *f = (int *)(uintptr_t)oa;
                out[9] = (double)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // completion tag for a host that polls the (pinned, coherent) result block instead of
            // waiting for the whole stream: results first, system-scope fence, then the tag
            __threadfence_system();
            out[8] = tag;
        }
    }
}

}  // namespace

// one thread per response cell up to 1024 cells (S <= 32), whole wavefronts
static inline unsigned decode_threads(int S)
{
    const int n = S * S;
    return (unsigned)(n >= 1024 ? 1024 : (n + 63) / 64 * 64);
}

extern "C" int usot_conf_fusion_reduce_f32(void *stream, const float *cv, float *out,
                                           int B, int M, int P, int C)
{
    if (!cv || !out || B <= 0 || M <= 0 || P <= 0 || C <= 0 || (C & 3)) return USOT_EINVAL;
    if (((uintptr_t)cv % 16) || ((uintptr_t)out % 16)) return USOT_EINVAL;
    const long total = (long)B * P * (C / 4);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL((conf_fusion_reduce_kernel<0, 0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const void *)cv, out, B, M, P, C / 4);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

/* the same reduction with the result stored as fp16 (out_dtype 1) or bf16 (2) and the input map cv in fp32 (in_dtype 0), fp16 (1)
 * or bf16 (2): fp32 arithmetic, one rounding */
extern "C" int usot_conf_fusion_reduce_lp(void *stream, const void *cv, int in_dtype, void *out, int B, int M, int P, int C, int out_dtype)
{
    if (!cv || !out || B <= 0 || M <= 0 || P <= 0 || C <= 0 || (C & 3) || (out_dtype != 1 && out_dtype != 2)) return USOT_EINVAL;
    if (in_dtype < 0 || in_dtype > 2 || ((uintptr_t)cv % (in_dtype ? 8 : 16)) || ((uintptr_t)out % 8)) return USOT_EINVAL;
    const long total = (long)B * P * (C / 4);
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
#define CF_LAUNCH(OT, IT) hipLaunchKernelGGL((conf_fusion_reduce_kernel<OT, IT>), dim3(blocks), dim3(256), 0, s, cv, (float *)out, B, M, P, C / 4)
    if (out_dtype == 1) { if (in_dtype == 0) CF_LAUNCH(1, 0); else if (in_dtype == 1) CF_LAUNCH(1, 1); else CF_LAUNCH(1, 2); }
    else                { if (in_dtype == 0) CF_LAUNCH(2, 0); else if (in_dtype == 1) CF_LAUNCH(2, 1); else CF_LAUNCH(2, 2); }
#undef CF_LAUNCH
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_prroi_pool_forward_f32(void *stream, const float *feat, const float *rois, float *out,
                                           int R, int C, int H, int W, int PH, int PW, float scale,
                                           int64_t f_sb, int64_t f_sc, int64_t f_sh, int64_t f_sw,
                                           int64_t o_sr, int64_t o_sc, int64_t o_sh, int64_t o_sw)
{
    if (R < 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0) return USOT_EINVAL;
    if (R == 0) return USOT_OK;      /* the reference binding early-returns on empty (…gpu.c:22-44) */
    if (!feat || !rois || !out) return USOT_EINVAL;
    PrK p{feat, rois, out, R, C, H, W, PH, PW, scale, f_sb, f_sc, f_sh, f_sw, o_sr, o_sc, o_sh, o_sw};
    const long total = (long)R * C * PH * PW;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(prroi_forward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

/* The reference's own native symbol with its own signature (prroi_pooling_gpu_impl.cuh:20-28,
 * .cu:387-402): contiguous NCHW features [B][C][H][W], rois [R][5], contiguous output
 * [R][C][PH][PW], top_count = R*C*PH*PW.  A binding compiled from the reference's
 * prroi_pooling_gpu.c:22-44 links against this.  The reference's launcher prints the runtime error
 * and calls exit(-1); this one prints the same kind of line and returns (a library must not end the
 * process); callers that want a status use usot_prroi_pool_forward_f32. */
extern "C" void PrRoIPoolingForwardGpu(hipStream_t stream, const float *bottom_data, const float *bottom_rois,
                                       float *top_data, const int channels_, const int height_, const int width_,
                                       const int pooled_height_, const int pooled_width_,
                                       const float spatial_scale_, const int top_count)
{
    const long per_roi = (long)channels_ * pooled_height_ * pooled_width_;
    if (per_roi <= 0 || top_count < 0 || top_count % per_roi) {
        fprintf(stderr, "PrRoIPoolingForwardGpu: invalid argument (top_count %d, C %d, PH %d, PW %d)\n",
                top_count, channels_, pooled_height_, pooled_width_);
        return;
    }
    const long HW = (long)height_ * width_, PP = (long)pooled_height_ * pooled_width_;
    const int rc = usot_prroi_pool_forward_f32((void *)stream, bottom_data, bottom_rois, top_data,
                                               (int)(top_count / per_roi), channels_, height_, width_,
                                               pooled_height_, pooled_width_, spatial_scale_,
                                               channels_ * HW, HW, width_, 1, channels_ * PP, PP, pooled_width_, 1);
    if (rc != USOT_OK) fprintf(stderr, "PrRoIPoolingForwardGpu: %s\n", usot_strerror(rc));
}

/* ---- the gradients of the same operator.  The reference's binding exposes them for training
 * (prroi_pooling_gpu.c:46-113 -> PrRoIPoolingBackwardGpu / PrRoIPoolingCoorBackwardGpu, .cu:404-440); contiguous
 * NCHW as there.  Both zero their output first, as the reference's launchers do (cudaMemsetAsync, .cu:415,436). */
extern "C" int usot_prroi_pool_backward_f32(void *stream, const float *rois, const float *top_diff, float *bottom_diff,
                                            int R, int B, int C, int H, int W, int PH, int PW, float scale)
{
    if (R < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || !bottom_diff) return USOT_EINVAL;
    if (hipMemsetAsync(bottom_diff, 0, sizeof(float) * (size_t)B * C * H * W, (hipStream_t)stream) != hipSuccess) return USOT_ELAUNCH;
    if (R == 0) return USOT_OK;
    if (!rois || !top_diff) return USOT_EINVAL;
    PrGradK p{nullptr, rois, nullptr, top_diff, bottom_diff, R, B, C, H, W, PH, PW, scale};
    const long total = (long)R * C * PH * PW;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(prroi_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_prroi_pool_coor_backward_f32(void *stream, const float *feat, const float *rois, const float *top_data,
                                                 const float *top_diff, float *rois_diff,
                                                 int R, int B, int C, int H, int W, int PH, int PW, float scale)
{
    if (R < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0) return USOT_EINVAL;
    if (R == 0) return USOT_OK;
    if (!feat || !rois || !top_data || !top_diff || !rois_diff || R > 65535) return USOT_EINVAL;
    if (hipMemsetAsync(rois_diff, 0, sizeof(float) * (size_t)R * 5, (hipStream_t)stream) != hipSuccess) return USOT_ELAUNCH;
    PrGradK p{feat, rois, top_data, top_diff, rois_diff, R, B, C, H, W, PH, PW, scale};
    const int per_roi = C * PH * PW;
    const int bx = (per_roi + 255) / 256 > 64 ? 64 : (per_roi + 255) / 256;
    hipLaunchKernelGGL(prroi_coor_backward_kernel, dim3(bx, R), dim3(256), 0, (hipStream_t)stream, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

/* exact signatures of prroi_pooling_gpu_impl.cuh:30-54; stderr + return instead of exit(-1), like the forward shim */
extern "C" void PrRoIPoolingBackwardGpu(hipStream_t stream, const float *bottom_data, const float *bottom_rois,
                                        const float *top_data, const float *top_diff, float *bottom_diff,
                                        const int channels_, const int height_, const int width_,
                                        const int pooled_height_, const int pooled_width_, const float spatial_scale_,
                                        const int top_count, const int bottom_count)
{
    (void)bottom_data; (void)top_data;                      /* unused by the feature gradient, there as here (.cu:416-418) */
    const long per_roi = (long)channels_ * pooled_height_ * pooled_width_, per_img = (long)channels_ * height_ * width_;
    if (per_roi <= 0 || per_img <= 0 || top_count < 0 || top_count % per_roi || bottom_count <= 0 || bottom_count % per_img) {
        fprintf(stderr, "PrRoIPoolingBackwardGpu: invalid argument (top_count %d, bottom_count %d)\n", top_count, bottom_count);
        return;
    }
    const int rc = usot_prroi_pool_backward_f32((void *)stream, bottom_rois, top_diff, bottom_diff, (int)(top_count / per_roi),
                                                (int)(bottom_count / per_img), channels_, height_, width_,
                                                pooled_height_, pooled_width_, spatial_scale_);
    if (rc != USOT_OK) fprintf(stderr, "PrRoIPoolingBackwardGpu: %s\n", usot_strerror(rc));
}

extern "C" void PrRoIPoolingCoorBackwardGpu(hipStream_t stream, const float *bottom_data, const float *bottom_rois,
                                            const float *top_data, const float *top_diff, float *bottom_diff,
                                            const int channels_, const int height_, const int width_,
                                            const int pooled_height_, const int pooled_width_, const float spatial_scale_,
                                            const int top_count, const int bottom_count)
{
    const long per_roi = (long)channels_ * pooled_height_ * pooled_width_;
    if (per_roi <= 0 || top_count < 0 || top_count % per_roi || bottom_count != (top_count / per_roi) * 5) {
        fprintf(stderr, "PrRoIPoolingCoorBackwardGpu: invalid argument (top_count %d, bottom_count %d)\n", top_count, bottom_count);
        return;
    }
    /* the signature carries no batch size (the reference never checks a RoI's batch index): accept any index >= 0 */
    const int rc = usot_prroi_pool_coor_backward_f32((void *)stream, bottom_data, bottom_rois, top_data, top_diff, bottom_diff,
                                                     (int)(top_count / per_roi), 0x7fffffff, channels_, height_, width_,
                                                     pooled_height_, pooled_width_, spatial_scale_);
    if (rc != USOT_OK) fprintf(stderr, "PrRoIPoolingCoorBackwardGpu: %s\n", usot_strerror(rc));
}

extern "C" int usot_permute4_f32(void *stream, const float *src, float *dst,
                                 int D0, int D1, int D2, int D3,
                                 int64_t s0, int64_t s1, int64_t s2, int64_t s3)
{
    if (!src || !dst || D0 <= 0 || D1 <= 0 || D2 <= 0 || D3 <= 0) return USOT_EINVAL;
    const long total = (long)D0 * D1 * D2 * D3;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(permute4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       src, dst, D1, D2, D3, total, (long)s0, (long)s1, (long)s2, (long)s3);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_decode_f32(void *stream, const float *cls, const float *cls_mem, const float *bbox,
                               const double *window, double *out, int S, int instance_size, int stride,
                               float ratio, double penalty_k, double window_influence,
                               double tw, double th)
{
    if (!cls || !cls_mem || !bbox || !window || !out || S <= 0 || S > 64) return USOT_EINVAL;
    hipLaunchKernelGGL(decode_kernel, dim3(1), dim3(decode_threads(S)), 0, (hipStream_t)stream, cls, cls_mem, bbox,
                       window, out, S, instance_size, stride, ratio, penalty_k, window_influence, tw, th,
                       (const double *)nullptr, (float *)nullptr);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

/* plan-side variant: target size read from device memory, PrPool roi of the best box written
 * to device memory, so a frame's decode + memory-feature pooling needs no host round trip */
extern "C" int usot_decode_dev_f32(void *stream, const float *cls, const float *cls_mem,
                                   const float *bbox, const double *window, double *out, int S,
                                   int instance_size, int stride, float ratio, double penalty_k,
                                   double window_influence, const double *tsz_dev, float *roi_out)
{
    if (!cls || !cls_mem || !bbox || !window || !out || !tsz_dev || S <= 0 || S > 64) return USOT_EINVAL;
    hipLaunchKernelGGL(decode_kernel, dim3(1), dim3(decode_threads(S)), 0, (hipStream_t)stream, cls, cls_mem, bbox,
                       window, out, S, instance_size, stride, ratio, penalty_k, window_influence, 1.0, 1.0,
                       tsz_dev, roi_out);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_crop_resize_u8_f32(void *stream, const unsigned char *im, float *out, int H, int W,
                                       int x0, int y0, int win, int S, int fill_b, int fill_g, int fill_r)
{
    if (!im || !out || H <= 0 || W <= 0 || win <= 0 || S <= 0) return USOT_EINVAL;
    CropK p{im, out, H, W, S, win, x0, y0, {fill_b, fill_g, fill_r}};
    hipLaunchKernelGGL(crop_resize_kernel, dim3((S * S + 255) / 256), dim3(256), 0, (hipStream_t)stream, p);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_rows_copy_f32(void *stream, const float *src, const int32_t *idx_dev, float *dst,
                                  int n_rows, int row_len, int scatter)
{
    if (!src || !idx_dev || !dst || n_rows <= 0 || row_len <= 0 || (row_len & 3)) return USOT_EINVAL;
    if (((uintptr_t)src % 16) || ((uintptr_t)dst % 16)) return USOT_EINVAL;
    const long total = (long)n_rows * (row_len / 4);
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(rows_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       src, (const int *)idx_dev, dst, n_rows, row_len / 4, scatter);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_rows_copy_multi_f32(void *stream, int nseg, const float *const *src, const int32_t *idx_dev,
                                        float *const *dst, int n_rows, const int32_t *row_len, int scatter,
                                        int32_t *stash_next)
{
    if (scatter && stash_next) return USOT_EINVAL;
    if (nseg < 1 || nseg > 4 || !src || !dst || !idx_dev || !row_len || n_rows <= 0) return USOT_EINVAL;
    RowsK k;
    long most = 0;
    for (int i = 0; i < 4; ++i) {
        const int q = i < nseg ? i : 0;
        if (!src[q] || !dst[q] || row_len[q] <= 0 || (row_len[q] & 3)) return USOT_EINVAL;
        if (((uintptr_t)src[q] % 16) || ((uintptr_t)dst[q] % 16)) return USOT_EINVAL;
        k.src[i] = src[q]; k.dst[i] = dst[q]; k.row_len4[i] = row_len[q] / 4;
        const long t = (long)n_rows * k.row_len4[i];
        if (t > most) most = t;
    }
    k.nseg = nseg; k.n_rows = n_rows; k.scatter = scatter; k.stash = stash_next;
    const int blocks = (int)((most + 255) / 256 > 1024 ? 1024 : (most + 255) / 256);
    hipLaunchKernelGGL(rows_copy_multi_kernel, dim3(blocks, nseg), dim3(256), 0, (hipStream_t)stream, k, (const int *)idx_dev);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_rows_append_gather_f32(void *stream, const float *const *fresh, float *const *bank, float *const *picked,
                                           const int32_t *row_len, const int32_t *idx_dev, int n_pick, int slot_pos)
{
    if (!fresh || !bank || !picked || !row_len || !idx_dev || n_pick < 1 || n_pick > 32 || slot_pos < 0) return USOT_EINVAL;
    RowsAG k;
    long most = 0;
    for (int i = 0; i < 4; ++i) {
        if (!fresh[i] || !bank[i] || row_len[i] <= 0 || (row_len[i] & 3)) return USOT_EINVAL;
        if (((uintptr_t)fresh[i] % 16) || ((uintptr_t)bank[i] % 16)) return USOT_EINVAL;
        if (i && (!picked[i - 1] || ((uintptr_t)picked[i - 1] % 16))) return USOT_EINVAL;
        k.fresh[i] = fresh[i]; k.bank[i] = bank[i]; k.row_len4[i] = row_len[i] / 4;
        if (i) k.picked[i - 1] = picked[i - 1];
        const long t = (long)(i ? n_pick : 1) * k.row_len4[i];
        if (t > most) most = t;
    }
    if (most > 0x3fffffffL) return USOT_EINVAL;
    k.n_pick = n_pick; k.slot_pos = slot_pos;
    const int blocks = (int)((most + 255) / 256 > 64 ? 64 : (most + 255) / 256);
    hipLaunchKernelGGL(rows_append_gather_kernel, dim3(blocks, 7), dim3(256), 0, (hipStream_t)stream, k, (const int *)idx_dev);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_thin_conv3x3_f32(void *stream, const usot_conv_desc *d, int n)
{
    if (!d || n < 1 || n > 4) return USOT_EINVAL;
    ThinK k;
    k.n = n; k.N = d[0].N; k.H = d[0].H; k.W = d[0].W; k.Cin = d[0].Cin;
    int total = 0;
    for (int i = 0; i < 4; ++i) {
        const usot_conv_desc &c = d[i < n ? i : 0];
        if (i < n) {
            if (!c.x || !c.w || !c.y || c.res || c.KH != 3 || c.KW != 3 || c.stride != 1 || c.pad_h != 1 || c.pad_w != 1 ||
                c.dil_h != 1 || c.dil_w != 1 || !c.y_nchw || c.ksplit > 1 || c.y_coff || c.act_split > 0) return USOT_EINVAL;
            if (c.N != k.N || c.H != k.H || c.W != k.W || c.Cin != k.Cin || c.OH != c.H || c.OW != c.W) return USOT_EINVAL;
            if (c.Cout < 1 || c.Cout > 4 || (c.Cin & 3) || c.N <= 0) return USOT_EINVAL;
            if (((uintptr_t)c.x % 16) || ((uintptr_t)c.w % 16) || ((c.x_gs | c.w_gs) & 3)) return USOT_EINVAL;
        }
        const int groups = c.groups > 1 ? c.groups : 1;
        k.x[i] = c.x; k.w[i] = c.w; k.bias[i] = c.bias; k.y[i] = c.y;
        k.cout[i] = c.Cout; k.act[i] = c.act; k.groups[i] = groups; k.ycs[i] = c.Cout; k.yco[i] = 0;
        k.x_gs[i] = c.x_gs; k.w_gs[i] = c.w_gs; k.y_gs[i] = c.y_gs; k.b_gs[i] = (int)c.b_gs;
        k.start[i] = total;
        if (i < n) total += groups * c.N * c.H * c.W;
    }
    for (int i = n; i < 5; ++i) k.start[i] = total;
    // batches: a wavefront per output ROW with the filters in registers (thin_row) once the rows alone fill the chip
    // (>= 1024 wavefronts: 8 streams of 25 rows x 6 problem-groups); one frame keeps the wavefront-per-pixel form
    long rows = 0;
    for (int i = 0; i < n; ++i) rows += (long)k.groups[i] * k.N * k.H;
    if (k.Cin == 256 && k.W <= 64 && rows >= 1024 && d[0].tile != 70) {          // (lane ow keeps pixel ow of the row)
        // 4-channel problems (bbox_pred) as two 2-channel ones: same arithmetic per output, the activations read twice (L2)
        for (int i = 0; i < n && k.n < 4; ++i)
            if (k.cout[i] == 4 && k.groups[i] == 1) {
                const int j = k.n++;
                k.x[j] = k.x[i]; k.w[j] = k.w[i] + 2L * 9 * k.Cin; k.bias[j] = k.bias[i] ? k.bias[i] + 2 : nullptr; k.y[j] = k.y[i];
                k.cout[i] = k.cout[j] = 2; k.act[j] = k.act[i]; k.groups[j] = 1;
                k.x_gs[j] = k.x_gs[i]; k.w_gs[j] = k.w_gs[i]; k.y_gs[j] = k.y_gs[i]; k.b_gs[j] = k.b_gs[i];
                k.ycs[i] = k.ycs[j] = 4; k.yco[i] = 0; k.yco[j] = 2;
            }
        n = k.n;
        int maxco = 1;
        for (int i = 0; i < n; ++i) maxco = k.cout[i] > maxco ? k.cout[i] : maxco;
        int tr = 0;
        for (int i = 0; i < 4; ++i) {
            k.start[i] = tr;
            if (i < n) tr += k.groups[i] * k.N * k.H;
        }
        k.start[4] = tr;
        for (int i = n; i < 4; ++i) k.start[i] = tr;
        if (maxco <= 2) hipLaunchKernelGGL(thin_conv3x3_rows_kernel<2>, dim3((unsigned)((tr + 3) / 4)), dim3(256), 0, (hipStream_t)stream, k);
        else            hipLaunchKernelGGL(thin_conv3x3_rows_kernel<4>, dim3((unsigned)((tr + 3) / 4)), dim3(256), 0, (hipStream_t)stream, k);
        USOT_CHECK_LAUNCH();
        return USOT_OK;
    }
    hipLaunchKernelGGL(thin_conv3x3_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, k);
    USOT_CHECK_LAUNCH();
    return USOT_OK;
}

extern "C" int usot_device_slot(void)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= USOT_MAX_DEV) return -1;
    return dev;
}

extern "C" int usot_device_guard(void) { return usot_device_slot() >= 0 ? USOT_OK : USOT_ESTATE; }

extern "C" int usot_abi_version(void) { return 6; }   // 2: usot_conv_desc.w_frag; 3: w_scale; 4: x_split / y_split; 5: ovf (conv + pw_pair descriptors), decode's out[9]; 6: usot_rows_append_gather_f32

extern "C" const char *usot_strerror(int code)
{
    switch (code) {
    case USOT_OK: return "ok";
    case USOT_EINVAL: return "invalid argument or unsupported geometry";
    case USOT_ELAUNCH: return "kernel launch failed";
    case USOT_ENOMEM: return "out of memory";
    case USOT_ESTATE: return "object used in the wrong state";
    case USOT_ENOTBUILT: return "conv tile not compiled into this library (experimental tile: build with USOT_EXPERIMENTS=1)";
    default: return "unknown error";
    }
}
