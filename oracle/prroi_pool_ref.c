/* TEST INFRASTRUCTURE — CPU oracle, never linked or called by the product path.
 *
 * Plain-C restatement of Precise RoI Pooling (forward, and the two gradients the reference's
 * binding exposes for training), following the reference's CUDA kernels:
 *   lib/models/prroi_pool/src/prroi_pooling_gpu_impl.cu
 *     :37-42   PrRoIPoolingGetData        (out-of-bounds taps read 0)
 *     :71-106  PrRoIPoolingMatCalculation (closed-form integral of one unit cell)
 *     :149-212 PrRoIPoolingForward        (bin loop, zero-area bins -> 0)
 *     :108-147 PrRoIPoolingMatDistributeDiff, :214-272 PrRoIPoolingBackward (feature gradient)
 *     :50-69   SingleCoorIntegral / Interpolation, :274-380 PrRoIPoolingCoorBackward (RoI gradient)
 * The reference has no CPU implementation (functional.py:62-63 raises) and no test
 * pins it: PARITY UNPINNED — anchored only by the analytic known answers in
 * tests/test_oracle_prpool.py.
 *
 * Arithmetic is float32 with the same operation order as the kernel so that a GPU
 * implementation following the same order can be compared tightly.
 * Layout: features [B][C][H][W] contiguous, rois [R][5] = (batch, x1, y1, x2, y2),
 * out [R][C][PH][PW].
 */
#include <math.h>

static float tap(const float *d, int h, int w, int H, int W)
{
    if (h < 0 || w < 0 || h >= H || w >= W) return 0.0f;
    return d[h * W + w];
}

/* integral over [x0,x1]x[y0,y1] (inside the unit cell with corner (s_h,s_w)) of the
 * bilinear interpolant; the four terms are the four cell corners. */
static float cell_integral(const float *d, int s_h, int s_w, int e_h, int e_w,
                           float y0, float x0, float y1, float x1, int H, int W)
{
    float alpha, beta, lim_alpha, lim_beta, tmp, sum = 0.0f;

    alpha = x0 - (float)s_w;  beta = y0 - (float)s_h;
    lim_alpha = x1 - (float)s_w;  lim_beta = y1 - (float)s_h;
    tmp = (lim_alpha - 0.5f * lim_alpha * lim_alpha - alpha + 0.5f * alpha * alpha)
        * (lim_beta - 0.5f * lim_beta * lim_beta - beta + 0.5f * beta * beta);
    sum += tap(d, s_h, s_w, H, W) * tmp;

    alpha = (float)e_w - x1;  lim_alpha = (float)e_w - x0;
    tmp = (lim_alpha - 0.5f * lim_alpha * lim_alpha - alpha + 0.5f * alpha * alpha)
        * (lim_beta - 0.5f * lim_beta * lim_beta - beta + 0.5f * beta * beta);
    sum += tap(d, s_h, e_w, H, W) * tmp;

    alpha = x0 - (float)s_w;  beta = (float)e_h - y1;
    lim_alpha = x1 - (float)s_w;  lim_beta = (float)e_h - y0;
    tmp = (lim_alpha - 0.5f * lim_alpha * lim_alpha - alpha + 0.5f * alpha * alpha)
        * (lim_beta - 0.5f * lim_beta * lim_beta - beta + 0.5f * beta * beta);
    sum += tap(d, e_h, s_w, H, W) * tmp;

    alpha = (float)e_w - x1;  lim_alpha = (float)e_w - x0;
    tmp = (lim_alpha - 0.5f * lim_alpha * lim_alpha - alpha + 0.5f * alpha * alpha)
        * (lim_beta - 0.5f * lim_beta * lim_beta - beta + 0.5f * beta * beta);
    sum += tap(d, e_h, e_w, H, W) * tmp;
    return sum;
}

void prroi_pool_forward_ref(const float *features, const float *rois, float *out,
                            int R, int C, int H, int W, int PH, int PW, float scale)
{
    for (int n = 0; n < R; ++n) {
        const float *roi = rois + n * 5;
        int b = (int)roi[0];
        float rsw = roi[1] * scale, rsh = roi[2] * scale;
        float rew = roi[3] * scale, reh = roi[4] * scale;
        float rw = fmaxf(rew - rsw, 0.0f), rh = fmaxf(reh - rsh, 0.0f);
        float bh = rh / (float)PH, bw = rw / (float)PW;
        float win_size = fmaxf(0.0f, bw * bh);
        for (int c = 0; c < C; ++c) {
            const float *d = features + ((long)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    float *o = out + (((long)n * C + c) * PH + ph) * PW + pw;
                    if (win_size == 0.0f) { *o = 0.0f; continue; }
                    float wsw = rsw + bw * pw, wsh = rsh + bh * ph;
                    float wew = wsw + bw, weh = wsh + bh;
                    int s_w = (int)floorf(wsw), e_w = (int)ceilf(wew);
                    int s_h = (int)floorf(wsh), e_h = (int)ceilf(weh);
                    float sum = 0.0f;
                    for (int wi = s_w; wi < e_w; ++wi)
                        for (int hi = s_h; hi < e_h; ++hi)
                            sum += cell_integral(d, hi, wi, hi + 1, wi + 1,
                                                 fmaxf(wsh, (float)hi), fmaxf(wsw, (float)wi),
                                                 fminf(weh, (float)hi + 1.0f),
                                                 fminf(wew, (float)wi + 1.0f), H, W);
                    *o = sum / win_size;
                }
        }
    }
}

/* ---- gradients.  The kernels accumulate with atomicAdd in an unspecified order; here the sums run
 * sequentially in index order (n, c, ph, pw; cells w-major as the kernel's loops), so agreement with a
 * GPU implementation is to float32 reassociation, not bit-exact. ------------------------------------ */

/* weight of a cell corner over [a, b] measured from that corner: int_a^b (1 - t) dt, the factor both
 * MatCalculation (.cu:79-80) and MatDistributeDiff (.cu:124-125) form per axis */
static float corner_w(float a, float b) { return b - 0.5f * b * b - a + 0.5f * a * a; }

static void add_at(float *diff, float v, int h, int w, int H, int W)          /* .cu:108-113 */
{
    if (h < 0 || w < 0 || h >= H || w >= W) return;
    diff[h * W + w] += v;
}

/* feature gradient (.cu:214-272): bottom_diff [B][C][H][W] is zeroed here as the launcher does (.cu:415) */
void prroi_pool_backward_ref(const float *rois, const float *top_diff, float *bottom_diff,
                             int R, int B, int C, int H, int W, int PH, int PW, float scale)
{
    for (long i = 0; i < (long)B * C * H * W; ++i) bottom_diff[i] = 0.0f;
    for (int n = 0; n < R; ++n) {
        const float *roi = rois + n * 5;
        int b = (int)roi[0];
        float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
        float bh = fmaxf(reh - rsh, 0.0f) / (float)PH, bw = fmaxf(rew - rsw, 0.0f) / (float)PW;
        float win_size = fmaxf(0.0f, bw * bh);
        for (int c = 0; c < C; ++c) {
            float *d = bottom_diff + ((long)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    float g = top_diff[(((long)n * C + c) * PH + ph) * PW + pw];
                    float share = win_size == 0.0f ? 0.0f : g / win_size;
                    float wsw = rsw + bw * pw, wsh = rsh + bh * ph, wew = wsw + bw, weh = wsh + bh;
                    int s_w = (int)floorf(wsw), e_w = (int)ceilf(wew), s_h = (int)floorf(wsh), e_h = (int)ceilf(weh);
                    for (int wi = s_w; wi < e_w; ++wi)
                        for (int hi = s_h; hi < e_h; ++hi) {
                            float x0 = fmaxf(wsw, (float)wi), x1 = fminf(wew, (float)wi + 1.0f);
                            float y0 = fmaxf(wsh, (float)hi), y1 = fminf(weh, (float)hi + 1.0f);
                            float xn = corner_w(x0 - (float)wi, x1 - (float)wi), xf = corner_w((float)(wi + 1) - x1, (float)(wi + 1) - x0);
                            float yn = corner_w(y0 - (float)hi, y1 - (float)hi), yf = corner_w((float)(hi + 1) - y1, (float)(hi + 1) - y0);
                            add_at(d, share * (xn * yn), hi, wi, H, W);
                            add_at(d, share * (xf * yn), hi, wi + 1, H, W);
                            add_at(d, share * (xn * yf), hi + 1, wi, H, W);
                            add_at(d, share * (xf * yf), hi + 1, wi + 1, H, W);
                        }
                }
        }
    }
}

static float lerp_weight(float dh, float dw) { return (1.0f - fabsf(dh)) * (1.0f - fabsf(dw)); }   /* .cu:44-48 */

static float surface(const float *d, float h, float w, int H, int W)                                  /* .cu:54-69 */
{
    int h1 = (int)floorf(h), w1 = (int)floorf(w);
    float v = tap(d, h1, w1, H, W) * lerp_weight(h - (float)h1, w - (float)w1);
    v += tap(d, h1 + 1, w1, H, W) * lerp_weight(h - (float)(h1 + 1), w - (float)w1);
    v += tap(d, h1, w1 + 1, H, W) * lerp_weight(h - (float)h1, w - (float)(w1 + 1));
    v += tap(d, h1 + 1, w1 + 1, H, W) * lerp_weight(h - (float)(h1 + 1), w - (float)(w1 + 1));
    return v;
}

/* integral over [s, t] of the line through c1 (at 0) and c2 (at 1)  (.cu:50-52) */
static float segment(float s, float t, float c1, float c2)
{
    return 0.5f * (t * t - s * s) * c2 + (t - 0.5f * t * t - s + 0.5f * s * s) * c1;
}

/* RoI gradient (.cu:274-380): rois_diff [R][5], zeroed here (.cu:436); column 0 stays 0 */
void prroi_pool_coor_backward_ref(const float *features, const float *rois, const float *top_data,
                                  const float *top_diff, float *rois_diff,
                                  int R, int C, int H, int W, int PH, int PW, float scale)
{
    for (int i = 0; i < R * 5; ++i) rois_diff[i] = 0.0f;
    for (int n = 0; n < R; ++n) {
        const float *roi = rois + n * 5;
        int b = (int)roi[0];
        float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
        float bh = fmaxf(reh - rsh, 0.0f) / (float)PH, bw = fmaxf(rew - rsw, 0.0f) / (float)PW;
        float win_size = fmaxf(0.0f, bw * bh);
        float *gr = rois_diff + n * 5;
        for (int c = 0; c < C; ++c) {
            const float *d = features + ((long)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    long idx = (((long)n * C + c) * PH + ph) * PW + pw;
                    float g = top_diff[idx], top = top_data[idx];
                    float share = win_size == 0.0f ? 0.0f : g / win_size;
                    if (share == 0.0f) continue;                                  /* .cu:315-317 */
                    float wsw = rsw + bw * pw, wsh = rsh + bh * ph, wew = wsw + bw, weh = wsh + bh;
                    int s_w = (int)floorf(wsw), e_w = (int)ceilf(wew), s_h = (int)floorf(wsh), e_h = (int)ceilf(weh);
                    float gx1 = 0.0f, gx2 = 0.0f, gy1 = 0.0f, gy2 = 0.0f;
                    for (int hi = s_h; hi < e_h; ++hi) {
                        float s = fmaxf(wsh, (float)hi) - hi, t = fminf(weh, (float)(hi + 1)) - hi;
                        gx1 += segment(s, t, surface(d, hi, wsw, H, W), surface(d, hi + 1, wsw, H, W));
                        gx2 += segment(s, t, surface(d, hi, wew, H, W), surface(d, hi + 1, wew, H, W));
                    }
                    for (int wi = s_w; wi < e_w; ++wi) {
                        float s = fmaxf(wsw, (float)wi) - wi, t = fminf(wew, (float)(wi + 1)) - wi;
                        gy1 += segment(s, t, surface(d, wsh, wi, H, W), surface(d, wsh, wi + 1, H, W));
                        gy2 += segment(s, t, surface(d, weh, wi, H, W), surface(d, weh, wi + 1, H, W));
                    }
                    float px1 = (-gx1 + (weh - wsh) * top) / win_size * scale;
                    float py1 = (-gy1 + (wew - wsw) * top) / win_size * scale;
                    float px2 = (gx2 - (weh - wsh) * top) / win_size * scale;
                    float py2 = (gy2 - (wew - wsw) * top) / win_size * scale;
                    gr[1] += (float)((px1 * (1.0 - (float)pw / PW) + px2 * (1.0 - (float)(pw + 1) / PW)) * g);
                    gr[2] += (float)((py1 * (1.0 - (float)ph / PH) + py2 * (1.0 - (float)(ph + 1) / PH)) * g);
                    gr[3] += (float)((px2 * (float)(pw + 1) / PW + px1 * (float)pw / PW) * g);
                    gr[4] += (float)((py2 * (float)(ph + 1) / PH + py1 * (float)ph / PH) * g);
                }
        }
    }
}
