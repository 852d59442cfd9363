/* usot_hip.h — C ABI of libusot_hip.so: hand-written gfx950 (MI355X) kernels for the
 * USOT Siamese-tracking forward pass.
 *
 * Conventions (mirroring the reference's only native ABI,
 * lib/models/prroi_pool/src/prroi_pooling_gpu_impl.cuh:20-28):
 *   - the caller owns every buffer; kernels borrow raw DEVICE pointers for the duration
 *     of the enqueue; nothing is allocated or retained by the library (plans excepted);
 *   - every call enqueues asynchronously on the given hipStream_t (passed as void*);
 *   - every entry point returns 0 on success or a negative USOT_E* code — never exit()
 *     (the reference's launcher calls exit(-1), prroi_pooling_gpu_impl.cu:20-27);
 *   - all tensors are float32 unless a name says otherwise; "NHWC" = channels innermost.
 *
 * Each entry point cites the reference code it replaces.
 */
#ifndef USOT_HIP_H
#define USOT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define USOT_OK            0
#define USOT_EINVAL       -1   /* bad shape / unsupported geometry */
#define USOT_ELAUNCH      -2   /* hipGetLastError() after launch   */
#define USOT_ENOMEM       -3
#define USOT_ESTATE       -4   /* plan used in the wrong state     */
#define USOT_ENOTBUILT    -5   /* a conv tile id that exists only in the -DUSOT_EXPERIMENTS build (usot_conv_tile_built) */

/* activation codes of the conv epilogue */
#define USOT_ACT_NONE      0
#define USOT_ACT_RELU      1
#define USOT_ACT_EXP       2   /* exp(x)                  connect.py:236-237 (bbox head)      */
#define USOT_ACT_CONF      3   /* exp(min(max(x,0),4))    connect.py:128-131 (ReLU then clamp) */

int usot_abi_version(void);
/* Launchers cache per-DEVICE state (zero pages, LDS-limit raises, CU counts) in tables indexed by the current HIP device:
 * usot_device_slot() = that index (hipGetDevice; 0 .. 15) or -1; usot_device_guard() = USOT_OK when the current device has a slot,
 * USOT_ESTATE otherwise.  One process may drive one GPU (the benchmarked form: one rank per GPU) or several (set the device, use its
 * streams); until round 6 the library bound itself to the first device it saw. */
int usot_device_slot(void);
int usot_device_guard(void);
/* HBM ceiling probe of the box (csrc/bw_probe.hip): mode 0 read `bytes`, 1 copy `bytes`, 2 read `bytes` + write bytes / 4
 * (GroupDW's byte mix, one interleaved read stream, non-temporal stores), 3 GroupDW's TRAFFIC without its compute - an address-level
 * emulation of the batched launch for S = bytes / (3 * 841 * 1024) samples: (sample, 64-channel group) blocks reading 256-byte
 * granules of three maps in src (S * 3 * 841 KiB) and writing S * 625 KiB to dst.  bytes % 4096 == 0 (mode 3: % 1024). */
int usot_bw_probe(void *stream, const void *src, void *dst, int64_t bytes, int mode);
const char *usot_strerror(int code);

/* ---- convolution as implicit GEMM on fp32 MFMA (v_mfma_f32_16x16x4_f32) -------------
 * y = act(conv(x, w) + bias [+ res]).  BatchNorm is folded into w/bias by the host.
 * Replaces every nn.Conv2d + nn.BatchNorm2d (+ReLU, +residual add) launch of
 * lib/models/modules.py:37-56,138-146 and lib/models/connect.py:19-53,110-119,178-215.
 *   x    NHWC  [N][H][W][Cin], Cin % 32 == 0
 *   w    packed [Cout][KH*KW*Cin], k = (kh*KW + kw)*Cin + ci
 *   y    NHWC  [N][OH][OW] with pixel stride y_cstride and channel offset y_coff, or
 *        NCHW  [N][Cout][OH][OW] when y_nchw != 0
 *   res  optional, laid out like an NHWC y with (res_cstride, res_coff)
 *   groups > 1 runs `groups` independent problems of identical geometry in one launch
 *   (pointer strides *_gs in elements): the three head towers, connect.py:178-207.
 *   act applies to channels [0, act_split) and act2 to [act_split, Cout) (act_split >= Cout
 *   means act everywhere): lets conf_gen|value_gen run as one Cout=512 conv.
 *   ksplit > 1 splits the K loop over `ksplit` workgroups; partials go to `ws` and the LAST slice of a
 *   tile to arrive sums them (in slice order) and applies the epilogue inside the same launch.  `ws` holds
 *   usot_conv_ws_floats(d) floats — the slabs [ksplit][groups][M][Cout] followed by one ticket word per tile —
 *   and must be ZERO before its first use (the kernel leaves the tickets zero again).
 *   tile: 0 = heuristic, else one of USOT_TILE_* ids (see usot_conv_tile_count).
 */
typedef struct usot_conv_desc {
    const float *x, *w, *bias, *res;
    float *y;
    float *ws;
    int32_t N, H, W, Cin, OH, OW, Cout;
    int32_t KH, KW, stride, pad_h, pad_w, dil_h, dil_w;
    int32_t y_cstride, y_coff, res_cstride, res_coff, y_nchw;
    int32_t act, act2, act_split;
    int32_t groups;
    int64_t x_gs, w_gs, b_gs, y_gs, r_gs;
    int32_t ksplit, tile;
    int32_t w_frag;   /* 1: `w` is in MFMA fragment order (usot_conv_pack_wfrag_f32) — required by, and only valid with, the
                       * weight-streaming tiles (usot_conv_tile_wfrag(tile) == 1); row offsets / group strides of the bank must
                       * be multiples of 16 rows */
    int32_t defer;    /* with ksplit > 1: 1 = DEFERRED reduction - the launch writes its ksplit partial tiles [ksplit][groups][M][Cout]
                       * to `ws` (plain stores, no tickets) and applies NEITHER bias NOR activation; `y` is not written.  The consumer
                       * sums the slabs, adds the bias and activates while it stages its input (usot_pw_pair_desc.t2_parts).  A kernel
                       * boundary is the synchronisation: no in-launch combine (4-6 us on the tail of a 20 us launch), no second launch */
    const float *w_scale;   /* w_frag == 2 (the split-fp16 tiles, usot_conv_tile_wfrag(tile) == 2): `w` holds every filter row as hi + lo
                       * fp16 of (row x a power of two) - per k-tile of 64: 64 hi halves then 64 lo halves, the 256 bytes of the fp32 row
                       * segment - and w_scale[groups][Cout] = 1 / (that power of two x 8) multiplies the finished sums (8 = the
                       * activation scale the kernel applies before it splits them).  NULL otherwise. */
    int32_t x_split;  /* 1: `x` is a SPLIT map - per pixel and 64-channel block the 64 hi halves then the 64 lo halves (fp16) of 8 x value,
                       * the 256 bytes of the fp32 block - as written by a launch with y_split = 1.  Required by, and only valid with, the
                       * all-DMA split-fp16 tiles (usot_conv_tile_xsplit(tile) == 1): both operands then reach LDS by LDS-DMA */
    int32_t y_split;  /* 1: the result is written as a split map (split-fp16 tiles only; dense NHWC, Cout % 64 == 0, no residual, no
                       * split-K).  Its only readers are launches with x_split = 1 */
    int32_t *ovf;     /* split-fp16 tiles: NULL, or a device word the launch sets to 1 (sticky: it never clears it) when one of its finished
                       * sums is not finite - what an activation beyond the fp16 window (|x| >= 8 188: hi = inf, lo = -inf) or a non-finite
                       * input turns every sum it enters into - BEFORE bias / activation / split-K slabs see the sum (a ReLU would turn the
                       * NaN into a finite 0).  The caller re-runs the work on the exact-fp32 tiles (usot_amd/engine.py: Session, Engine.track) */
} usot_conv_desc;

int usot_conv2d_f32(void *stream, const usot_conv_desc *d);
/* up to 4 convolutions of different geometry side by side in ONE launch (same tile shape =
 * d[0].tile, own ksplit each): shortcut conv + conv1 of a bottleneck, the three dilated
 * encoders of one input, the prediction heads. */
int usot_conv2d_batch_f32(void *stream, const usot_conv_desc *d, int n);
/* thin 3x3 / stride 1 / pad 1 convolutions with 1-4 output channels and NCHW output (the prediction
 * heads, connect.py:236-241,275): up to four descriptors of the same input geometry per launch, one
 * wavefront per output pixel; honours x, w, bias, y, N, H, W, Cin, Cout, act, groups and the group
 * strides of the descriptor.  From 1024 output rows (Cin = 256: batches of >= 14 streams) the launcher
 * switches to one wavefront per output ROW with the filters in registers and a sliding 3x3 window —
 * bit-identical results; d[0].tile = 70 forces the per-pixel form (tests). */
int usot_thin_conv3x3_f32(void *stream, const usot_conv_desc *d, int n);
int usot_plan_add_thin_conv(void *plan, const usot_conv_desc *d, int n);
int usot_conv_tile_count(void);
int usot_conv_tile_info(int tile, int *bm, int *bn);           /* tile ids are 1..count */
int usot_conv_tile_built(int tile);                            /* 1: compiled into this library (the routed tiles; every id with -DUSOT_EXPERIMENTS) */
int usot_experiments_built(void);                              /* 1: the library was built with -DUSOT_EXPERIMENTS */
int usot_conv_tile_name(int tile, char *buf, int len);         /* kernel symbol of the tile */
int usot_conv_tile_xsplit(int tile);                           /* 1: the tile reads a split input map (usot_conv_desc.x_split) */
int usot_conv_tile_wfrag(int tile);                            /* 1: the tile streams its filters in fragment order; 2: split-fp16 bank + w_scale */
/* weight-stationary tiles (filters held in registers, k split over the 8 waves of a workgroup) serve ONE K each: returns it
 * (0: the tile takes any K); *kpanel = the multiple Cin must have (128 / 256).  They also need Cout % 32 == 0, ksplit == 1. */
int usot_conv_tile_kreq(int tile, int *kpanel);
/* persistent stream-K tiles (conv_igemm_f32_v3p): the launch is ONE resident set of workgroups that share the (tile, k-tile)
 * units of all its problems evenly; shares that end inside a tile meet through write-through slabs and per-wave tickets in
 * d[0].ws (usot_conv_streamk_ws_floats floats, zero before first use).  ksplit must be 1. */
int usot_conv_tile_streamk(int tile);
int64_t usot_conv_streamk_ws_floats(const usot_conv_desc *d, int n, int tile);
/* filter bank [Cout][K] (K % 64 == 0) -> ceil(Cout/16)*16*K floats in fragment order
 * [16-row block][k-tile of 64][round of 16 k][lane][4 k]; rows past Cout zero.  One launch, any stream. */
int usot_conv_pack_wfrag_f32(void *stream, const float *w, float *wf, int Cout, int K);
int64_t usot_conv_ws_floats(const usot_conv_desc *d);          /* workspace need for ksplit */

/* ---- bf16 variant for the batched backbone (BASELINE config 3): x / w / res / y are bf16
 * NHWC (uint16 storage), bias fp32, accumulate fp32 on v_mfma_f32_16x16x32_bf16.  Uses
 * N..dil_w, act (NONE | RELU), res, tile of the descriptor; dense output; Cin % 64 == 0,
 * Cout % 4 == 0.                                                                         */
int usot_conv2d_bf16(void *stream, const usot_conv_desc *d);
/* general low-precision form: dtype 0 = bf16, 1 = fp16 (v_mfma_f32_16x16x32_f16); out_f32 != 0
 * stores the result as fp32 (outputs feeding the fp32 xcorr / reduce / prediction convs, BASELINE
 * config 5).  Also honoured: act / act2 / act_split (NONE, RELU, EXP, CONF) and groups with x_gs,
 * w_gs, b_gs, y_gs (y_gs in OUTPUT elements; no residual with groups).  Not supported: ksplit,
 * y_nchw, channel-offset outputs.  tile: 0 = heuristic, 1..usot_conv_bf16_tile_count().           */
int usot_conv2d_lp(void *stream, const usot_conv_desc *d, int dtype, int out_f32);
int usot_conv_bf16_tile_built(int tile);                       /* 1: the low-precision tile id is compiled into this library (cf. usot_conv_tile_built) */
int usot_cvt_f32_to_lp(void *stream, const float *src, void *dst, int64_t n, int dtype);
int usot_maxpool3x3s2_lp(void *stream, const void *x, void *y, int N, int H, int W, int C, int OH, int OW, int dtype);
int usot_conv_bf16_tile_count(void);
/* fused low-precision stem (7x7/s2 conv + BN + ReLU) + 3x3/s2/p1 max-pool on the bf16|fp16 MFMA:
 * modules.py:70-74,138-141 in one launch.  wfrag: filter bank as MFMA A fragments [4][6][64][8].
 * The kernel convolves x - mu[ci]; bias must already contain sum_k w[co][k] * mu[ci(k)].
 * dtype 0: bf16 fragments, crop as hi + lo bf16 (two MFMAs per fragment), bf16 output; 1: fp16 throughout; 2: fp16
 * fragments and crop (11 significant bits on both operands, one MFMA per fragment), bf16 OUTPUT — the stem of the bf16
 * backbone (|x - mu| < 152 and BN-folded stem filters are far inside the fp16 range). */
int usot_stem_pool_lp(void *stream, const float *x, const void *wfrag, const float *bias, void *y,
                      int N, int H, int W, int OH, int OW, int PH, int PW, int dtype,
                      float mu0, float mu1, float mu2);
int usot_cvt_f32_to_bf16(void *stream, const float *src, void *dst, int64_t n);
int usot_maxpool3x3s2_bf16(void *stream, const void *x, void *y, int N, int H, int W, int C, int OH, int OW);

/* ---- fused pair of pointwise convolutions of the low-precision backbone: a bottleneck's conv3 + BN + residual +
 * ReLU (modules.py:48-56) and the NEXT block's conv1 + BN + ReLU (modules.py:40-42; or the neck's 1x1, connect.py:
 * 294-300) in one persistent launch — the 4x-wide map Y is written once and never read back, the next tile's reads fly
 * under the second GEMM (csrc/pw_pair.hip).  All activations bf16|fp16 NHWC dense, fp32 accumulate, biases fp32:
 *   Y[M][CO] = relu(t2[M][CM] . w3^T + b3 + res[M][CO]);   T[M][CN] = act2(Y . w1^T + b1)
 * w3p and w1 are in the kernel's FRAGMENT order (every MFMA fragment load = one contiguous KiB): build them with
 * usot_pw_pair_layout, which lists for each packed 16-byte chunk the source (row, first k); b3 and b1 natural order.
 * Shapes: usot_pw_pair_supported(CM, CO, CN).  act2: USOT_ACT_NONE | USOT_ACT_RELU.  dtype 0 = bf16, 1 = fp16.     */
typedef struct usot_pw_pair_desc {
    const void *t2, *w3p, *res, *w1;
    const float *b3, *b1;
    void *y, *t;
    int32_t M, CM, CO, CN, act2;
    void *ws;             /* fp32 form only: usot_pw_pair_f32_ws_floats() zero-initialised floats, or NULL */
    /* fp32 form only: t2 given as t2_parts > 1 partial sums [t2_parts][M][CM] of the producing convolution (usot_conv_desc.defer);
     * the kernel stages relu(sum of the parts in order + t2_bias) as its pixel tile.  0 / 1: t2 is the finished map. */
    int32_t t2_parts, res_parts;
    const float *t2_bias;
    /* fp32 form only: res given as res_parts > 1 partial sums [res_parts][M][CO] of the shortcut convolution (defer); the kernel
     * adds (sum of the parts in order + res_bias) as the residual - no activation: the downsample branch has none */
    const float *res_bias;
    int32_t *ovf;         /* usot_pw_pair_f32s only: NULL or the sticky not-finite word of usot_conv_desc.ovf (checked on both GEMMs' sums) */
} usot_pw_pair_desc;
int usot_pw_pair_lp(void *stream, const usot_pw_pair_desc *d, int dtype);
int usot_pw_pair_layout(int CM, int CO, int CN, int which, int32_t *row, int32_t *k0);   /* which: 0 = w3, 1 = w1 */
int usot_pw_pair_supported(int CM, int CO, int CN);
int usot_plan_add_pw_pair(void *plan, const usot_pw_pair_desc *d, int dtype);   /* dtype 2: the fp32 form below */

/* one pointwise EXPANSION convolution of the batched low-precision backbone, pixel-stationary (csrc/pw_panel.hip: the
 * panel's X fragments in registers, W streamed through LDS once per 256 pixels, register epilogue in 32-byte pieces):
 * y[M][N] = act(x[M][K] . w^T + bias (+ res)); x, w ([N][K], the conv kernels' layout), res, y in the storage type
 * (dtype 0 = bf16, 1 = fp16), bias fp32 or NULL, res NULL or [M][N], act USOT_ACT_NONE | USOT_ACT_RELU.
 * Replaces the 1x1 convs of modules.py:48-56,108-113 at batch 64.  Shapes: usot_pw_panel_supported(K, N). */
int usot_pw_panel_lp(void *stream, const void *x, const void *w, const float *bias, const void *res, void *y,
                     int M, int K, int N, int act, int dtype);
int usot_pw_panel_supported(int K, int N);
int usot_pw_panel_pixels(int CM, int CO, int CN);      /* pixels per panel = per workgroup (CN = 0: single conv); 0 = unsupported */
int usot_pw_panel_min_pixels(int K, int N);           /* the smallest panel of the single convolution (used below 192 default panels) */
int usot_plan_add_pw_panel(void *plan, const void *x, const void *w, const float *bias, const void *res, void *y,
                           int M, int K, int N, int act, int dtype);
/* ... and the fused pair of usot_pw_pair_lp in that form: Y's sixteen channels per lane, rounded, are the B fragments of
 * the second GEMM (no LDS image of Y, no re-read from HBM).  Descriptor as for usot_pw_pair_lp EXCEPT that both banks
 * are in the conv kernels' natural layout: d->w3p = w3 [CO][CM], d->w1 = w1 [CN][CO]. */
int usot_pw_panel_pair_lp(void *stream, const usot_pw_pair_desc *d, int dtype);
int usot_pw_panel_pair_supported(int CM, int CO, int CN);
int usot_plan_add_pw_panel_pair(void *plan, const usot_pw_pair_desc *d, int dtype);

/* A bottleneck's conv2 -> conv3 of the batched low-precision backbone in ONE launch (csrc/conv_pw_lp.hip; modules.py:43-56):
 * y[M][4 CM] = relu(relu(conv(x; c2->w) + c2->bias) . w3^T + b3 + res[M][4 CM]).  c2 describes conv2 (x, w [CM][KH*KW*CM] in the
 * conv kernels' layout, bias, N, H, W, OH, OW, KH, KW, stride, pad, dil; Cin = Cout = CM, act = USOT_ACT_RELU; its y is IGNORED:
 * a panel of conv2's output stays in LDS and is consumed there), w3 [4 CM][CM], b3 fp32, res / y dense; storage type dtype 0 =
 * bf16, 1 = fp16.  Bit-identical to usot_conv2d_lp followed by usot_pw_panel_lp.  Shapes: usot_conv_pw_supported(Cin, CM, CO):
 * (256, 256, 1024) layer3, (128, 128, 512) layer2.  A workgroup owns a panel of usot_conv_pw_pixels(M) pixels: 256 (16
 * wavefronts), or 128 (8 wavefronts) when M gives fewer than 192 panels of 256 or a mostly empty second round of them (CUs < panels
 * < 1.5 CUs); c2->tile & 3 = 1 / 2 forces the 256 / 128 form.
 * 3 x 3 / stride 1 / pad = dil <= 4 convolutions run their k-loop ROW-SHARED (one staged activation tile per (kh, channel chunk)
 * serves the three kw taps; k order (kh, chunk, kw): same products, another fp32 summation order); c2->tile & 4 keeps the per-tap
 * loop, whose result is bit-identical to the unfused launches. */
int usot_conv_pw_lp(void *stream, const usot_conv_desc *c2, const void *w3, const float *b3, const void *res, void *y, int dtype);
int usot_conv_pw_supported(int Cin, int CM, int CO);
int usot_conv_pw_pixels(int64_t M);
/* ... and the pair form: the NEXT block's conv1 rides along as in usot_pw_panel_pair_lp, T[M][CN] = act2(Y . w1^T + b1) - a whole
 * bottleneck tail (conv2 -> conv3 + residual + ReLU -> next conv1) per launch.  d as for usot_pw_panel_pair_lp (w3p = w3 [CO][CM],
 * w1 [CN][CO], natural layouts; M = conv2's output pixels) except that d->t2 is ignored.  Bit-identical to usot_conv2d_lp followed
 * by usot_pw_panel_pair_lp.  Shapes: usot_conv_pw_pair_supported(CM, CO, CN): (128, 512, 128), layer2 - and (256, 1024, 256), layer3,
 * where the second convolution runs as a FIFTH phase instead (no registers for the pair form): the workgroup reads its own Y panel
 * back (L2 / Infinity Cache) and computes T as a 16-k-tile implicit GEMM on the freed LDS - T bit-identical to usot_conv2d_lp on Y. */
int usot_conv_pw_pair_lp(void *stream, const usot_conv_desc *c2, const usot_pw_pair_desc *d, int dtype);
int usot_conv_pw_pair_supported(int CM, int CO, int CN);
int usot_plan_add_conv_pw_pair(void *plan, const usot_conv_desc *c2, const usot_pw_pair_desc *d, int dtype);
int usot_plan_add_conv_pw(void *plan, const usot_conv_desc *c2, const void *w3, const float *b3, const void *res, void *y, int dtype);
/* usot_conv_pw_pair_lp for layer3's blocks (CM = 256, CO = 1024, CN = 256; conv2 3 x 3 / stride 1 / pad = dil in 1..4) with the matrix-pipe
 * work and the HBM work OVERLAPPED on every CU (csrc/conv_pw_ov.hip; modules.py:43-56,40-42): two kinds of 8-wave workgroup share a CU -
 * one runs conv2 on 128-pixel panels and, later, the next conv1 on the panel's Y rows; the other takes the conv2 panels (through L2), runs
 * conv3 + residual + ReLU and streams Y out - paired through write-through stores and flags in `ws`.  Same arithmetic as the pair form with
 * the row-shared k-loop (k order (kh, 32-channel chunk, kw)); T bit-identical to usot_conv2d_lp on this launch's Y.
 * ws: usot_conv_pw_ov_ws_bytes(M) bytes (M = conv2's output pixels), 16-byte aligned, ZERO before the first launch; every launch leaves the
 * hand-off flags zero again (graph-replay safe).  The word after the 2 ceil(M / 128) flags is a sticky error flag: non-zero = a bounded
 * hand-off wait ran out (the launch finished on garbage instead of hanging). */
int usot_conv_pw_ov_lp(void *stream, const usot_conv_desc *c2, const usot_pw_pair_desc *d, int dtype, void *ws);
int usot_conv_pw_ov_supported(int CM, int CO, int CN);
int64_t usot_conv_pw_ov_ws_bytes(int64_t M);
int usot_plan_add_conv_pw_ov(void *plan, const usot_conv_desc *c2, const usot_pw_pair_desc *d, int dtype, void *ws);
int usot_conv_pw_ov_trace(void *buf);      /* measurement: placement + phase time stamps of later launches into buf (32 int64 per workgroup); NULL = off */

/* 3x3 / stride 1 / pad 1 convolution of the batched low-precision backbone as a direct convolution from an LDS halo tile
 * (csrc/conv3x3_halo.hip; layer1's conv2 + BN + ReLU, modules.py:43-46): x, y NHWC dense [N][H][W][C] and w [Cout][9 Cin]
 * (k = (kh*3 + kw)*Cin + ci, the conv kernels' layout) in the storage type (dtype 0 = bf16, 1 = fp16), bias fp32 or NULL,
 * act USOT_ACT_NONE | USOT_ACT_RELU.  Shapes: usot_conv3x3_halo_supported(Cin, Cout) (64 -> 64). */
int usot_conv3x3_halo_lp(void *stream, const void *x, const void *w, const float *bias, void *y,
                         int N, int H, int W, int Cin, int Cout, int act, int dtype);
int usot_conv3x3_halo_supported(int Cin, int Cout);
int usot_plan_add_conv3x3_halo(void *plan, const void *x, const void *w, const float *bias, void *y,
                               int N, int H, int W, int Cin, int Cout, int act, int dtype);

/* Layer1's FIRST bottleneck of the batched low-precision backbone plus the next block's conv1 in ONE launch
 * (csrc/bneck_lp.hip; modules.py:37-58 with the 1x1 downsample of modules.py:108-113): the 64-channel intermediates stay in
 * LDS, the shortcut conv is the second half of conv3's k axis.  NHWC dense, storage type dtype 0 = bf16 | 1 = fp16:
 * x [N][H][W][64]; w1 [64][64]; w2 [64][576] (k = (kh*3 + kw)*64 + ci); w3c [256][128] = [conv3 | downsample] along k;
 * wn [64][256] (the next block's conv1); biases fp32 (BN folded; b3c = conv3's + the downsample's);
 * y [N][H][W][256] = relu(conv3(relu(conv2(relu(conv1 x)))) + downsample(x)), t [N][H][W][64] = relu(conv1'(y)).
 * At least eight 8 x 16 tiles (N * ceil(H/8) * ceil(W/16)).  Shapes: usot_bneck_first_supported(Cin, Cmid, Cout, Cnext). */
typedef struct usot_bneck_desc {
    const void *x, *w1, *w2, *w3c, *wn;
    const float *b1, *b2, *b3c, *bn;
    void *y, *t;
    int32_t N, H, W;
} usot_bneck_desc;
int usot_bneck_first_lp(void *stream, const usot_bneck_desc *d, int dtype);
int usot_bneck_first_supported(int Cin, int Cmid, int Cout, int Cnext);
int usot_plan_add_bneck_first(void *plan, const usot_bneck_desc *d, int dtype);
/* The REST of a layer1 bottleneck whose conv1 output already exists (the launch above, or this one, made it) + the next block's
 * conv1, one launch: conv2 3x3 + BN + ReLU, conv3 1x1 + BN + identity residual + ReLU (modules.py:43-58), then conv1' + BN + ReLU
 * of the following block (modules.py:40-42).  The descriptor's fields are read as: x = t1 [N][H][W][64] (this block's conv1
 * output), w1 = the RESIDUAL map [N][H][W][256] (the block's input), w2 [64][576], w3c = w3 [256][64], wn [Cnext][256],
 * b2, b3c (= conv3's bias), bn; b1 unused; y [N][H][W][256], t [N][H][W][Cnext].  Cnext 64 | 128. */
int usot_bneck_tail_lp(void *stream, const usot_bneck_desc *d, int Cnext, int dtype);
int usot_bneck_tail_supported(int Cmid, int Cout, int Cnext);
int usot_plan_add_bneck_tail(void *plan, const usot_bneck_desc *d, int Cnext, int dtype);

/* Channel-reducing 1x1 convolution of the batched low-precision backbone with the accumulators stationary and K streaming
 * (csrc/pw_kstream.hip; layer3's conv1 + BN + ReLU 1024 -> 256, modules.py:40-42, and the neck's 1x1 + BN, connect.py:294-300):
 * y[M][N] = act(x[M][K] . w^T + bias), x / w ([N][K]) / y in the storage type (dtype 0 = bf16, 1 = fp16), bias fp32 or NULL,
 * act USOT_ACT_NONE | USOT_ACT_RELU.  Shapes: usot_pw_kstream_supported(K, N) ((1024, 256)). */
int usot_pw_kstream_lp(void *stream, const void *x, const void *w, const float *bias, void *y, long M, int K, int N, int act, int dtype);
int usot_pw_kstream_supported(int K, int N);
int usot_plan_add_pw_kstream(void *plan, const void *x, const void *w, const float *bias, void *y, long M, int K, int N, int act, int dtype);

/* 3x3 convolutions with K = 9 Cin >= 2304 of the batched low-precision backbone in the same accumulator-stationary form
 * (csrc/conv_kstream.hip; layer3's shortcut conv and conv2, layer2's shortcut conv, modules.py:43-46,115-126): a lane is an
 * output pixel and fetches its MFMA B fragments straight from global memory three k-chunks ahead, W streams through LDS.
 * x, y NHWC dense, w [Cout][9 Cin] (k = (kh*3 + kw)*Cin + ci) in the storage type (dtype 0 = bf16, 1 = fp16), bias fp32 or
 * NULL, stride 1 | 2, pad <= dil, act USOT_ACT_NONE | USOT_ACT_RELU.  Shapes: usot_conv_kstream_supported (Cin 256 | 512,
 * Cout a multiple of 256, 3 x 3). */
int usot_conv_kstream_lp(void *stream, const void *x, const void *w, const float *bias, void *y,
                         int N, int H, int W, int Cin, int Cout, int stride, int pad, int dil, int act, int dtype);
int usot_conv_kstream_supported(int Cin, int Cout, int KH, int KW);
int usot_plan_add_conv_kstream(void *plan, const void *x, const void *w, const float *bias, void *y,
                               int N, int H, int W, int Cin, int Cout, int stride, int pad, int dil, int act, int dtype);

/* the same pair in fp32 for the batch-1 frame (csrc/smallm_f32.hip; v_mfma_f32_16x16x4_f32, 16 pixels per workgroup):
 * every pointer of the descriptor is float32.  w3p / w1 in fragment order: the float at
 * [((cb * (K / 16) + r) * 64 + lane) * 4 + c] is W[cb * 16 + (lane & 15)][16 * r + 4 * (lane >> 4) + c]
 * (W = the [rows][K] filter bank: K = CM for w3p, CO for w1).  Shapes: usot_pw_pair_f32_supported.
 * With few pixel tiles (layer2 at batch 1-2) S = 4 workgroups share a tile, each owning a quarter of CO, and meet in
 * d->ws (write-through slabs + a ticket per tile, as the in-launch split-K of the conv kernels): size from
 * usot_pw_pair_f32_ws_floats (0 = not needed), zero before the first launch, owned by the caller; NULL = unsliced. */
int usot_pw_pair_f32(void *stream, const usot_pw_pair_desc *d);
int usot_pw_pair_f32_supported(int CM, int CO, int CN);
/* The same pair on SPLIT-fp16 operands (csrc/smallm_f32.hip: GemmRing BLO > 0; the arithmetic of the split-fp16 conv tiles): d->w3p and
 * d->w1 are the banks pre-split by the host - per column block of 16 rows and 32-k step a hi fragment (lane (quad, row) = 8 halves of
 * row x 2^e, k = 32 step + 8 quad ..) then the lo fragment, followed by the bank's per-row factors 1 / (2^e x 8) as floats - and the
 * kernel splits the staged pixel tile and its Y tile itself (x 8).  dtype 3 of usot_plan_add_pw_pair.  Shape (256, 1024, 256), sliced. */
int usot_pw_pair_f32s(void *stream, const usot_pw_pair_desc *d);
int usot_pw_pair_f32s_supported(int CM, int CO, int CN);
int64_t usot_pw_pair_f32_ws_floats(int M, int CM, int CO, int CN);

/* one pointwise convolution of the fp32 frame in the same style (16 pixels per workgroup, the pixel tile in LDS, filters
 * streamed from L2 in the fragment order above; several workgroups per pixel tile, each owning output channels):
 * y[M][N] = act(x[M][K] . W^T + b (+ res)), all float32 dense, act USOT_ACT_NONE | USOT_ACT_RELU, res may be NULL.   */
int usot_pw_single_f32(void *stream, const float *x, const float *wp, const float *b, const float *res, float *y,
                       int M, int K, int N, int act);
int usot_pw_single_f32_supported(int K, int N);
/* a whole bottleneck tail in one launch (layer1 of the fp32 frame): conv2 (3x3 / stride 1) + BN + ReLU on the im2col pixel
 * tile, then the pair of usot_pw_pair_f32 on its output without leaving the CU.  x NHWC [Nb][H][W][Cin] (conv1's output);
 * w2p conv2's packed bank [CM][9 Cin] in fragment order; d as for usot_pw_pair_f32 (d->t2 and d->ws ignored,
 * d->M = Nb * OH * OW).                                                                                               */
int usot_pw_triple_f32(void *stream, const float *x, const float *w2p, const float *b2, const usot_pw_pair_desc *d,
                       int Nb, int H, int W, int Cin, int OH, int OW, int pad_h, int pad_w, int dil_h, int dil_w);
int usot_pw_triple_f32_supported(int Cin, int CM, int CO, int CN);
int usot_plan_add_pw_triple(void *plan, const float *x, const float *w2p, const float *b2, const usot_pw_pair_desc *d,
                            int Nb, int H, int W, int Cin, int OH, int OW, int pad_h, int pad_w, int dil_h, int dil_w);

/* 3x3 / stride-1 convolution in that form (the pixel tile is the 16 x 9 Cin im2col image, zeros for padding taps):
 * x NHWC [Nb][H][W][Cin] dense, wp the packed bank [N][9 Cin] in fragment order, y [Nb * OH * OW][N] (+ res of that shape). */
int usot_stream_conv3x3_f32(void *stream, const float *x, const float *wp, const float *b, const float *res, float *y,
                            int Nb, int H, int W, int Cin, int OH, int OW, int N, int pad_h, int pad_w, int dil_h, int dil_w,
                            int act);
int usot_stream_conv3x3_f32_supported(int Cin, int N);
int usot_plan_add_stream_conv3x3(void *plan, const float *x, const float *wp, const float *b, const float *res, float *y,
                                 int Nb, int H, int W, int Cin, int OH, int OW, int N, int pad_h, int pad_w, int dil_h, int dil_w,
                                 int act);
int usot_plan_add_pw_single(void *plan, const float *x, const float *wp, const float *b, const float *res, float *y,
                            int M, int K, int N, int act);

/* ---- stem: 7x7 / stride 2 / pad 0 conv, 3 -> 64 channels, + folded BN + ReLU ---------
 * modules.py:70-72,138-140.  x NCHW [N][3][H][W] (the API-edge crop, BGR 0..255),
 * w packed [147][64] with row = (ci*7 + kh)*7 + kw, y NHWC [N][OH][OW][64].            */
int usot_stem_conv_f32(void *stream, const float *x, const float *w, const float *bias,
                       float *y, int N, int H, int W, int OH, int OW);
/* the same on x - mu[ci] (exact for this pad-0 conv; `bias` must then carry + sum_k w[k][co] * mu[ci(k)], folded in
 * float64 by the caller): raw crops sit around ~100, and the 147-tap float32 chain of every output otherwise rides on
 * 100 * sum(w).  mu = 0 is usot_stem_conv_f32. */
int usot_stem_conv_mu_f32(void *stream, const float *x, const float *w, const float *bias,
                          float *y, int N, int H, int W, int OH, int OW, float mu0, float mu1, float mu2);

/* ---- max-pool 3x3 / stride 2 / pad 1 on NHWC (modules.py:74,141) ------------------- */
int usot_maxpool3x3s2_f32(void *stream, const float *x, float *y,
                          int N, int H, int W, int C, int OH, int OW);

/* ---- depthwise cross-correlation, NCHW planes (drop-in for xcorr_depthwise,
 * lib/models/connect.py:147-157): out[p][i][j] = sum_uv x[p][i+u][j+v] * k[p][u][v]
 * for P = B*C planes; one wavefront per plane pair, template tile in LDS, window taps
 * exchanged between lanes with DPP/shuffles.  Wx <= 64.                                */
int usot_xcorr_depthwise_f32(void *stream, const float *x, const float *k, float *out,
                             int P, int Hx, int Wx, int Hk, int Wk);

/* ---- fused GroupDW on NHWC (connect.py:86-102): three depthwise xcorrs and the
 * softmax(weight)-weighted sum in one pass, no intermediate maps.
 *   branch b: x_b NHWC [XS][OH+hk_b-1][OW+wk_b-1] (pixel stride x_cs, channel offset x_co)
 *             z_b NHWC [S][hk_b][wk_b]            (pixel stride z_cs, channel offset z_co)
 *   sample s uses search map s / x_rep (x_rep = N_q for the memory branch, where the
 *   reference materialises a 7x repeat, connect.py:258-264) and template s.
 *   out NHWC [S][OH][OW][C];  wsm = softmax(weight) (3 floats, host pointer).          */
typedef struct usot_groupdw_desc {
    const float *x[3];
    const float *z[3];
    float *out;
    int32_t hk[3], wk[3];
    int32_t x_cs[3], x_co[3], z_cs[3], z_co[3];
    float wsm[3];
    int32_t S, x_rep, OH, OW, C;
    int32_t cols_per_thread;     /* kernel variant: 0 auto (strips for a frame; from 64 samples up the LDS-DMA
                                  * kernel for 25- and 27-wide responses, else ring);
                                  * 1: 5x1 strips; 50: 5x5 patches; 2: column threads;
                                  * 3: LDS row streaming; 4: ring (taps in LDS, rows streamed once);
                                  * 6: LDS-DMA (loader waves feed row-sets to LDS, 7-column strip waves) */
} usot_groupdw_desc;
int usot_groupdw_f32(void *stream, const usot_groupdw_desc *d);
/* the cols_per_thread code the launcher resolves 0 (auto) to for this many samples (benchmarks name
 * the kernel they time with it) */
int usot_groupdw_auto_variant(int total_samples, int OW);
/* up to three segments of identical geometry (the cls, reg and memory GroupDWs of a frame)
 * in ONE launch */
int usot_groupdw_multi_f32(void *stream, const usot_groupdw_desc *d, int nseg);
/* ... with the OUTPUT maps stored as fp16 (out_dtype 1) or bf16 (2) — d[i].out then points to 16-bit elements; the same fp32
 * arithmetic, one rounding at the store (the batched mixed-precision heads of BASELINE configs[4] feed these maps to fp16
 * convolutions).  25- and 27-wide responses only. */
int usot_groupdw_multi_lp(void *stream, const usot_groupdw_desc *d, int nseg, int out_dtype);

/* ---- Conf_Fusion reduction (connect.py:132-142): cv NHWC [B*M][P][2C] holding
 * conf = exp(clamp) in channels [0,C) and value in [C,2C) -> out [B][P][C] =
 * sum_m conf*value / sum_m conf.                                                       */
int usot_conf_fusion_reduce_f32(void *stream, const float *cv, float *out,
                                int B, int M, int P, int C);
int usot_conf_fusion_reduce_lp(void *stream, const void *cv, int in_dtype, void *out, int B, int M, int P, int C, int out_dtype);   /* cv fp32 (0) | fp16 (1) | bf16 (2); out fp16 (1) | bf16 (2) */

/* ---- Precise RoI Pooling forward.  Replaces PrRoIPoolingForwardGpu
 * (prroi_pooling_gpu_impl.cuh:20-28 / .cu:149-212,387-402) with explicit strides so the
 * same kernel reads NCHW (API edge) or NHWC (engine) features and writes either layout.
 *   feat element (b,c,h,w) at b*f_sb + c*f_sc + h*f_sh + w*f_sw
 *   rois [R][5] = (batch, x1, y1, x2, y2);  out element (r,c,ph,pw) at
 *   r*o_sr + c*o_sc + ph*o_sh + pw*o_sw                                                */
int usot_prroi_pool_forward_f32(void *stream, const float *feat, const float *rois, float *out,
                                int R, int C, int H, int W, int PH, int PW, float scale,
                                int64_t f_sb, int64_t f_sc, int64_t f_sh, int64_t f_sw,
                                int64_t o_sr, int64_t o_sc, int64_t o_sh, int64_t o_sw);

/* ---- the reference's native symbol, with the reference's signature
 * (lib/models/prroi_pool/src/prroi_pooling_gpu_impl.cuh:20-28; launcher .cu:387-402): contiguous
 * NCHW float32 features, rois [R][5], contiguous [R][C][PH][PW] output, top_count = R*C*PH*PW,
 * enqueued on `stream`.  prroi_pooling_gpu.c:22-44 links against it unchanged.
 * A shim over usot_prroi_pool_forward_f32; on a bad argument or a failed launch it prints one line
 * to stderr and returns — it does not exit(-1) as the reference's launcher does
 * (prroi_pooling_gpu_impl.cu:20-27).                                                          */
typedef struct ihipStream_t *hipStream_t;      /* identical to <hip/hip_runtime_api.h>'s typedef */
void PrRoIPoolingForwardGpu(hipStream_t stream, const float *bottom_data, const float *bottom_rois,
                            float *top_data, const int channels_, const int height_, const int width_,
                            const int pooled_height_, const int pooled_width_,
                            const float spatial_scale_, const int top_count);

/* ---- gradients of Precise RoI Pooling: what the reference's binding exposes for training
 * (prroi_pooling_gpu.c:46-113).  Contiguous NCHW features [B][C][H][W], rois [R][5], top_data / top_diff
 * [R][C][PH][PW].  Both zero-fill their output on the stream first (as .cu:415,436).
 *   backward:       bottom_diff [B][C][H][W] += top_diff / bin area * (weight of the pixel in the bin integral)
 *   coor_backward:  rois_diff [R][5] = d sum(top_diff * out) / d (batch, x1, y1, x2, y2); column 0 is 0.
 *                   A RoI whose batch index is outside [0, B) contributes nothing.                              */
int usot_prroi_pool_backward_f32(void *stream, const float *rois, const float *top_diff, float *bottom_diff,
                                 int R, int B, int C, int H, int W, int PH, int PW, float scale);
int usot_prroi_pool_coor_backward_f32(void *stream, const float *feat, const float *rois, const float *top_data,
                                      const float *top_diff, float *rois_diff,
                                      int R, int B, int C, int H, int W, int PH, int PW, float scale);

/* the reference's symbols for them, exact signatures (prroi_pooling_gpu_impl.cuh:30-54; launchers .cu:404-440);
 * on a bad argument or a failed launch: one line on stderr and return, never exit(-1) */
void PrRoIPoolingBackwardGpu(hipStream_t stream, const float *bottom_data, const float *bottom_rois,
                             const float *top_data, const float *top_diff, float *bottom_diff,
                             const int channels_, const int height_, const int width_,
                             const int pooled_height_, const int pooled_width_, const float spatial_scale_,
                             const int top_count, const int bottom_count);
void PrRoIPoolingCoorBackwardGpu(hipStream_t stream, const float *bottom_data, const float *bottom_rois,
                                 const float *top_data, const float *top_diff, float *bottom_diff,
                                 const int channels_, const int height_, const int width_,
                                 const int pooled_height_, const int pooled_width_, const float spatial_scale_,
                                 const int top_count, const int bottom_count);

/* ---- layout changes at the API edge: generic 4-D strided copy ----------------------
 * dst[n][a][b][c] (dense) = src[n*s0 + a*s1 + b*s2 + c*s3]                              */
int usot_permute4_f32(void *stream, const float *src, float *dst,
                      int D0, int D1, int D2, int D3,
                      int64_t s0, int64_t s1, int64_t s2, int64_t s3);

/* ---- on-device decode of one frame (usot_tracker.py:138-163): float32 sigmoid and
 * offline/online blend, then (in float64, as numpy promotes there because the grids are
 * float64) box decode, size/ratio penalty, cosine window and first-max argmax.
 * cls/cls_mem [S*S] logits, bbox [4][S*S], window [S*S] float64;
 * out[8] = {best_index, blended_score, penalty, x1, y1, x2, y2, pscore} (float64).
 * tw/th = target size already multiplied by scale_z (usot_tracker.py:258).              */
int usot_decode_f32(void *stream, const float *cls, const float *cls_mem, const float *bbox,
                    const double *window, double *out, int S, int instance_size, int stride,
                    float ratio, double penalty_k, double window_influence,
                    double tw, double th);

/* same, with the target size read from device memory (double[2]) and, if roi_out != NULL,
 * the PrRoIPool box of the winning cell (usot_tracker.py:196, 329-350) written to
 * roi_out[5] = (0, x1, y1, x2, y2) in feature coordinates: no host round trip between
 * decode and memory-feature pooling.  tsz_dev[6] is a caller-chosen frame tag that is copied
 * to out[8] AFTER the results (system-scope fence in between), so `out` needs 9 doubles and a
 * host may poll out[8] in pinned memory instead of synchronising the stream.
 * tsz_dev[3], read as a 64-bit integer: 0, or the device address of the frame's sticky split-fp16 range word
 * (usot_conv_desc.ovf); its value is then published as out[9] (10 doubles) with the results and the word is cleared. */
int usot_decode_dev_f32(void *stream, const float *cls, const float *cls_mem, const float *bbox,
                        const double *window, double *out, int S, int instance_size, int stride,
                        float ratio, double penalty_k, double window_influence,
                        const double *tsz_dev, float *roi_out);

/* ---- SiamFC crop on the device (lib/utils/track_utils.py:30-119, get_subwindow_tracking):
 * window [x0, x0+win) x [y0, y0+win) of an HWC uint8 BGR frame (coordinates may leave the
 * image: mean-colour fill, values already truncated to uint8 like numpy's assignment),
 * OpenCV-style fixed-point bilinear resize to S x S when win != S, CHW float32 output.     */
int usot_crop_resize_u8_f32(void *stream, const unsigned char *im, float *out, int H, int W,
                            int x0, int y0, int win, int S, int fill_b, int fill_g, int fill_r);

/* ---- row gather (scatter = 0: dst[i] = src[idx[i]]) / scatter (dst[idx[i]] = src[i]) of
 * `n_rows` rows of `row_len` floats with indices read from DEVICE memory: selects the
 * memory-queue kernels of a frame (usot_tracker.py:222-256) and appends the new one
 * (:264) inside a captured graph.                                                        */
int usot_rows_copy_f32(void *stream, const float *src, const int32_t *idx_dev, float *dst,
                       int n_rows, int row_len, int scatter);

/* ---- launch plans: record the per-frame kernel sequence once, replay it natively, or
 * capture it into a hipGraph (one hipGraphLaunch per frame).  Pointers are baked at add
 * time, so they must refer to buffers that outlive the plan (the engine's workspace).
 * fork(lane>0): that lane waits for lane 0, following ops go to `lane`; fork(0): back to
 * lane 0 without waiting; join(lane): lane 0 waits for `lane`.  Lanes are extra streams
 * (parallel graph branches after capture); run() on an uncaptured plan ignores lanes and
 * issues everything in program order on the caller's stream.                             */
void *usot_plan_create(void);
void usot_plan_destroy(void *plan);
int usot_plan_size(void *plan);
int usot_plan_add_conv(void *plan, const usot_conv_desc *d);
int usot_plan_add_conv_batch(void *plan, const usot_conv_desc *d, int n);
int usot_plan_add_conv_bf16(void *plan, const usot_conv_desc *d);
int usot_plan_add_conv_lp(void *plan, const usot_conv_desc *d, int dtype, int out_f32);
int usot_plan_add_cvt_lp(void *plan, const float *src, void *dst, int64_t n, int dtype);
int usot_plan_add_maxpool_lp(void *plan, const void *x, void *y, int N, int H, int W, int C, int OH, int OW, int dtype);
int usot_plan_add_stem_pool_lp(void *plan, const float *x, const void *wfrag, const float *bias, void *y,
                               int N, int H, int W, int OH, int OW, int PH, int PW, int dtype,
                      float mu0, float mu1, float mu2);
int usot_plan_add_cvt_bf16(void *plan, const float *src, void *dst, int64_t n);
int usot_plan_add_maxpool_bf16(void *plan, const void *x, void *y, int N, int H, int W, int C, int OH, int OW);
int usot_plan_add_groupdw(void *plan, const usot_groupdw_desc *d);
int usot_plan_add_groupdw_multi(void *plan, const usot_groupdw_desc *d, int nseg);
int usot_plan_add_groupdw_multi_lp(void *plan, const usot_groupdw_desc *d, int nseg, int out_dtype);
/* fused fp32 stem + max-pool on the fp32 MFMA (the 125x125 stem map never reaches HBM); wfrag from
 * usot_amd/engine.py: pack_stem_f32 */
int usot_stem_pool_f32(void *stream, const float *x, const float *wfrag, const float *bias, float *y,
                       int N, int H, int W, int OH, int OW, int PH, int PW);
int usot_plan_add_stem_pool(void *plan, const float *x, const float *wfrag, const float *bias, float *y,
                            int N, int H, int W, int OH, int OW, int PH, int PW);
/* the same with the crop's address taken from DEVICE memory at run time: xptr_dev[0..1] = low / high half of the
 * address (0 = use x).  A captured frame can then read any resident crop without a copy into a baked input
 * buffer; the session's first kernel stashes the address there from the host's control block. */
int usot_stem_pool_ind_f32(void *stream, const float *x, const float *wfrag, const float *bias, float *y,
                           int N, int H, int W, int OH, int OW, int PH, int PW, const int32_t *xptr_dev);
int usot_plan_add_stem_pool_ind(void *plan, const float *x, const float *wfrag, const float *bias, float *y,
                                int N, int H, int W, int OH, int OW, int PH, int PW, const int32_t *xptr_dev);
int usot_plan_add_stem(void *plan, const float *x, const float *w, const float *bias, float *y,
                       int N, int H, int W, int OH, int OW);
/* ... and both stem forms on x - mu[ci] (see usot_stem_conv_mu_f32) */
int usot_stem_pool_mu_f32(void *stream, const float *x, const float *wfrag, const float *bias, float *y,
                          int N, int H, int W, int OH, int OW, int PH, int PW, const int32_t *xptr_dev,
                          float mu0, float mu1, float mu2);
int usot_plan_add_stem_pool_mu(void *plan, const float *x, const float *wfrag, const float *bias, float *y,
                               int N, int H, int W, int OH, int OW, int PH, int PW, const int32_t *xptr_dev,
                               float mu0, float mu1, float mu2);
int usot_plan_add_stem_mu(void *plan, const float *x, const float *w, const float *bias, float *y,
                          int N, int H, int W, int OH, int OW, float mu0, float mu1, float mu2);
int usot_plan_add_maxpool(void *plan, const float *x, float *y, int N, int H, int W, int C,
                          int OH, int OW);
int usot_plan_add_conf_reduce(void *plan, const float *cv, float *out, int B, int M, int P, int C);
int usot_plan_add_conf_reduce_lp(void *plan, const void *cv, int in_dtype, void *out, int B, int M, int P, int C, int out_dtype);
int usot_plan_add_prroi(void *plan, const float *feat, const float *rois, float *out,
                        int R, int C, int H, int W, int PH, int PW, float scale,
                        int64_t f_sb, int64_t f_sc, int64_t f_sh, int64_t f_sw,
                        int64_t o_sr, int64_t o_sc, int64_t o_sh, int64_t o_sw);
int usot_plan_add_permute(void *plan, const float *src, float *dst, int D0, int D1, int D2, int D3,
                          int64_t s0, int64_t s1, int64_t s2, int64_t s3);
int usot_plan_add_decode(void *plan, const float *cls, const float *cls_mem, const float *bbox,
                         const double *window, double *out, int S, int instance_size, int stride,
                         float ratio, double penalty_k, double window_influence,
                         const double *tsz_dev, float *roi_out);
/* the same for up to four banks of different row length (one launch): the session's raw memory
 * features and their three cached encodings share the row indices.  stash_next (gather only, may
 * be NULL): idx_dev[n_rows .. n_rows+2] are copied to stash_next[0..2] (the append row, and two free words the
 * session uses for the crop address of usot_plan_add_stem_pool_ind), so that a later scatter of the same frame
 * can take its row from device memory while the host already rewrites the control block. */
int usot_rows_copy_multi_f32(void *stream, int nseg, const float *const *src, const int32_t *idx_dev,
                             float *const *dst, int n_rows, const int32_t *row_len, int scatter,
                             int32_t *stash_next);
int usot_plan_add_rows_copy_multi(void *plan, int nseg, const float *const *src, const int32_t *idx_dev,
                                  float *const *dst, int n_rows, const int32_t *row_len, int scatter,
                             int32_t *stash_next);
/* append + gather in one launch (the session's 'defer_append' = 2 frame): fresh[0..3] = the memory feature the previous frame
 * pooled and its three kernel-side encodings (one row each), bank[0..3] their banks, picked[0..2] the frame's picked-kernel
 * buffers of banks 1-3 (n_pick <= 32 rows each).  Row idx_dev[slot_pos] of every bank receives the fresh row, rows
 * idx_dev[0 .. n_pick) of banks 1-3 are gathered - one that IS the appended row is taken from `fresh`.  idx_dev may be pinned
 * host memory (the control block).  Replaces usot_rows_copy_multi_f32 (scatter) + usot_rows_copy_multi_f32 (gather);
 * the reference keeps the queue in Python lists (usot_tracker.py:222-264). */
int usot_rows_append_gather_f32(void *stream, const float *const *fresh, float *const *bank, float *const *picked,
                                const int32_t *row_len, const int32_t *idx_dev, int n_pick, int slot_pos);
int usot_plan_add_rows_append_gather(void *plan, const float *const *fresh, float *const *bank, float *const *picked,
                                     const int32_t *row_len, const int32_t *idx_dev, int n_pick, int slot_pos);
int usot_plan_add_rows_copy(void *plan, const float *src, const int32_t *idx_dev, float *dst,
                            int n_rows, int row_len, int scatter);
int usot_plan_fork(void *plan, int lane);
int usot_plan_join(void *plan, int lane);
int usot_plan_capture(void *plan, void *stream);
int usot_plan_run(void *plan, void *stream);
/* per-op mean milliseconds per launch (HIP events on `stream`, eager, program order, each op
 * launched `reps` times between its two events), blocking */
int usot_plan_profile(void *plan, void *stream, int frames, int reps, float *ms_per_op);
/* info[4] = {kind, conv tile id, ksplit, groups} of op i (kind: 0 conv, 1 stem, 2 maxpool,
 * 3 groupdw, 4 conf_reduce, 5 prroi, 6 permute, 7 decode, 8 fork, 9 join)                */
int usot_plan_op_info(void *plan, int i, int *info);
int usot_conv_resolve_tile(const usot_conv_desc *d);

#ifdef __cplusplus
}
#endif
#endif /* USOT_HIP_H */
