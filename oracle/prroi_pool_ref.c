/* TEST INFRASTRUCTURE — CPU oracle, never linked or called by the product path.
 *
 * Plain-C restatement of Precise RoI Pooling, forward only, following the
 * reference's CUDA kernel:
 *   lib/models/prroi_pool/src/prroi_pooling_gpu_impl.cu
 *     :37-42   PrRoIPoolingGetData        (out-of-bounds taps read 0)
 *     :71-106  PrRoIPoolingMatCalculation (closed-form integral of one unit cell)
 *     :149-212 PrRoIPoolingForward        (bin loop, zero-area bins -> 0)
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
