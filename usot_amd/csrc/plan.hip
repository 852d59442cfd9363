// Launch plans: the native per-frame runtime.
//
// A tracked frame is ~70 dependent kernel launches of a few microseconds each; issued one
// by one from Python the host, not the GPU, sets the frame rate.  A plan records the whole
// sequence once (descriptors with baked device pointers into the engine's static
// workspace), replays it from C++ with no per-launch Python, and can be captured into a
// hipGraph so that a frame costs one hipGraphLaunch.  Independent branches of the head
// (the cls / reg / memory chains, shortcut convs of the backbone) may be placed on side
// "lanes": extra streams forked from and joined to the main stream with events, which
// capture turns into parallel graph branches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <new>
#include <vector>
#include "usot_hip.h"
#include "common.h"

extern "C" int usot_decode_dev_f32(void *stream, const float *cls, const float *cls_mem,
                                   const float *bbox, const double *window, double *out, int S,
                                   int instance_size, int stride, float ratio, double penalty_k,
                                   double window_influence, const double *tsz_dev, float *roi_out);

extern "C" int usot_conv_resolve_tile(const usot_conv_desc *d);

namespace {

enum Kind { K_CONV, K_STEM, K_POOL, K_GDW, K_CONF, K_PRROI, K_PERM, K_DECODE, K_FORK, K_JOIN, K_ROWS, K_CONVB, K_CVTB, K_POOLB, K_STEMB, K_ROWSM, K_THIN, K_STEMP, K_PWPAIR, K_PW1, K_SC3, K_PW3, K_PANEL, K_PANELP, K_HALO, K_KSTREAM, K_CKSTREAM, K_BNECK1, K_BNECKT, K_CONVPW, K_CONVPWP, K_CONVPWO, K_ROWSAG };

constexpr int kLanes = 4;     // lane 0 is the caller's stream

struct Op {
    Kind kind;
    int lane;
    usot_conv_desc conv;          // first (or only) problem
    usot_conv_desc more[3];       // further problems of a batched launch
    int nconv;
    usot_groupdw_desc gdw[3];
    int ngdw;
    usot_pw_pair_desc pw;
    usot_bneck_desc bneck;
    const void *p[6];
    int i[8];
    int64_t l[8];
    float f[4];
    double d[2];
};

struct Plan {
    std::vector<Op> ops;
    int cur_lane = 0;
    hipStream_t side[kLanes] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> events;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool captured = false;
};

int issue(Plan *pl, hipStream_t main_stream, bool lanes, hipEvent_t *marks = nullptr, int reps = 1)
{
    hipStream_t st[kLanes];
    st[0] = main_stream;
    for (int k = 1; k < kLanes; ++k) st[k] = lanes ? pl->side[k] : main_stream;
    size_t ev = 0;
    size_t opi = 0;
    for (const Op &op : pl->ops) {
        hipStream_t s = st[op.lane];
        int rc = USOT_OK;
        if (marks && hipEventRecord(marks[opi], s) != hipSuccess) return USOT_ELAUNCH;
        ++opi;
        for (int rep = 0; rep < reps && rc == USOT_OK; ++rep)
        switch (op.kind) {
        case K_CONV:
            if (op.nconv <= 1) rc = usot_conv2d_f32(s, &op.conv);
            else {
                usot_conv_desc tmp[4];
                tmp[0] = op.conv;
                for (int q = 1; q < op.nconv; ++q) tmp[q] = op.more[q - 1];
                rc = usot_conv2d_batch_f32(s, tmp, op.nconv);
            }
            break;
        case K_GDW:  rc = op.i[6] ? usot_groupdw_multi_lp(s, op.gdw, op.ngdw, op.i[6]) : usot_groupdw_multi_f32(s, op.gdw, op.ngdw); break;
        case K_STEM:
            rc = usot_stem_conv_mu_f32(s, (const float *)op.p[0], (const float *)op.p[1], (const float *)op.p[2],
                                       (float *)op.p[3], op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.f[1], op.f[2], op.f[3]);
            break;
        case K_POOL:
            rc = usot_maxpool3x3s2_f32(s, (const float *)op.p[0], (float *)op.p[1], op.i[0], op.i[1], op.i[2],
                                       op.i[3], op.i[4], op.i[5]);
            break;
        case K_CONF:
            rc = op.i[6] ? usot_conf_fusion_reduce_lp(s, op.p[0], op.i[5], (void *)op.p[1], op.i[0], op.i[1], op.i[2], op.i[3], op.i[6])
                         : usot_conf_fusion_reduce_f32(s, (const float *)op.p[0], (float *)op.p[1], op.i[0], op.i[1],
                                                       op.i[2], op.i[3]);
            break;
        case K_PRROI:
            rc = usot_prroi_pool_forward_f32(s, (const float *)op.p[0], (const float *)op.p[1], (float *)op.p[2],
                                             op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.f[0],
                                             op.l[0], op.l[1], op.l[2], op.l[3], op.l[4], op.l[5], op.l[6], op.l[7]);
            break;
        case K_PERM:
            rc = usot_permute4_f32(s, (const float *)op.p[0], (float *)op.p[1], op.i[0], op.i[1], op.i[2], op.i[3],
                                   op.l[0], op.l[1], op.l[2], op.l[3]);
            break;
        case K_DECODE:
            rc = usot_decode_dev_f32(s, (const float *)op.p[0], (const float *)op.p[1], (const float *)op.p[2],
                                     (const double *)op.p[3], (double *)op.p[4], op.i[0], op.i[1], op.i[2],
                                     op.f[0], op.d[0], op.d[1], (const double *)op.p[5], (float *)op.l[0]);
            break;
        case K_CONVB: rc = usot_conv2d_lp(s, &op.conv, op.i[6], op.i[7]); break;
        case K_PANEL:
            rc = usot_pw_panel_lp(s, op.p[0], op.p[1], (const float *)op.p[2], op.p[3], (void *)op.p[4], op.i[0], op.i[1], op.i[2],
                                  op.i[3], op.i[6]);
            break;
        case K_PANELP: rc = usot_pw_panel_pair_lp(s, &op.pw, op.i[6]); break;
        case K_CONVPWP: rc = usot_conv_pw_pair_lp(s, &op.conv, &op.pw, op.i[6]); break;
        case K_CONVPWO: rc = usot_conv_pw_ov_lp(s, &op.conv, &op.pw, op.i[6], (void *)op.p[0]); break;
        case K_CONVPW: rc = usot_conv_pw_lp(s, &op.conv, op.p[0], (const float *)op.p[1], op.p[2], (void *)op.p[3], op.i[6]); break;
        case K_HALO:
            rc = usot_conv3x3_halo_lp(s, op.p[0], op.p[1], (const float *)op.p[2], (void *)op.p[3], op.i[0], op.i[1], op.i[2], op.i[3],
                                      op.i[4], op.i[5], op.i[6]);
            break;
        case K_BNECK1: rc = usot_bneck_first_lp(s, &op.bneck, op.i[6]); break;
        case K_BNECKT: rc = usot_bneck_tail_lp(s, &op.bneck, op.i[5], op.i[6]); break;
        case K_CKSTREAM:
            rc = usot_conv_kstream_lp(s, op.p[0], op.p[1], (const float *)op.p[2], (void *)op.p[3], op.i[0], op.i[1], op.i[2], op.i[3],
                                      op.i[4], op.i[5], op.i[7], (int)op.l[0], (int)op.l[1], op.i[6]);
            break;
        case K_KSTREAM:
            rc = usot_pw_kstream_lp(s, op.p[0], op.p[1], (const float *)op.p[2], (void *)op.p[3], op.l[0], op.i[1], op.i[2], op.i[3], op.i[6]);
            break;
        case K_PWPAIR: rc = op.i[6] == 3 ? usot_pw_pair_f32s(s, &op.pw) : op.i[6] == 2 ? usot_pw_pair_f32(s, &op.pw) : usot_pw_pair_lp(s, &op.pw, op.i[6]); break;
        case K_PW3:
            rc = usot_pw_triple_f32(s, (const float *)op.p[0], (const float *)op.p[1], (const float *)op.p[2], &op.pw,
                                    op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6], op.i[7], (int)op.l[0], (int)op.l[1]);
            break;
        case K_SC3:
            rc = usot_stream_conv3x3_f32(s, (const float *)op.p[0], (const float *)op.p[1], (const float *)op.p[2], (const float *)op.p[3],
                                         (float *)op.p[4], op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6], op.i[7],
                                         (int)op.l[0], (int)op.l[1], (int)op.l[2], (int)op.l[3]);
            break;
        case K_PW1:
            rc = usot_pw_single_f32(s, (const float *)op.p[0], (const float *)op.p[1], (const float *)op.p[2], (const float *)op.p[3],
                                    (float *)op.p[4], op.i[0], op.i[1], op.i[2], op.i[3]);
            break;
        case K_CVTB:  rc = usot_cvt_f32_to_lp(s, (const float *)op.p[0], (void *)op.p[1], op.l[0], op.i[6]); break;
        case K_POOLB:
            rc = usot_maxpool3x3s2_lp(s, op.p[0], (void *)op.p[1], op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6]);
            break;
        case K_STEMP:
            rc = usot_stem_pool_mu_f32(s, (const float *)op.p[0], (const float *)op.p[1], (const float *)op.p[2], (float *)op.p[3],
                                       op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6], (const int32_t *)op.p[4],
                                       op.f[1], op.f[2], op.f[3]);
            break;
        case K_THIN: {
            usot_conv_desc tmp[4];
            tmp[0] = op.conv;
            for (int q = 1; q < op.nconv; ++q) tmp[q] = op.more[q - 1];
            rc = usot_thin_conv3x3_f32(s, tmp, op.nconv);
            break;
        }
        case K_ROWSM: {
            const float *srcs[4] = {(const float *)op.p[0], (const float *)op.p[1], (const float *)op.p[2], (const float *)op.p[3]};
            float *dsts[4] = {(float *)op.l[0], (float *)op.l[1], (float *)op.l[2], (float *)op.l[3]};
            rc = usot_rows_copy_multi_f32(s, op.i[0], srcs, (const int32_t *)op.p[4], dsts, op.i[1], &op.i[3], op.i[2],
                                          (int32_t *)op.p[5]);
            break;
        }
        case K_ROWSAG: {
            const float *fresh[4] = {(const float *)op.p[0], (const float *)op.p[1], (const float *)op.p[2], (const float *)op.p[3]};
            float *bank[4] = {(float *)op.l[0], (float *)op.l[1], (float *)op.l[2], (float *)op.l[3]};
            float *picked[3] = {(float *)op.l[4], (float *)op.l[5], (float *)op.l[6]};
            rc = usot_rows_append_gather_f32(s, fresh, bank, picked, &op.i[2], (const int32_t *)op.p[4], op.i[0], op.i[1]);
            break;
        }
        case K_STEMB:
            rc = usot_stem_pool_lp(s, (const float *)op.p[0], op.p[1], (const float *)op.p[2], (void *)op.p[3], op.i[0], op.i[1],
                                   op.i[2], op.i[3], op.i[4], op.i[5], op.i[7], op.i[6], op.f[1], op.f[2], op.f[3]);
            break;
        case K_ROWS:
            rc = usot_rows_copy_f32(s, (const float *)op.p[0], (const int32_t *)op.p[1], (float *)op.p[2],
                                    op.i[0], op.i[1], op.i[2]);
            break;
        case K_FORK:      // lane op.lane waits for everything issued so far on lane 0
            if (lanes && op.lane != 0) {
                if (ev >= pl->events.size()) return USOT_ESTATE;
                hipEvent_t e = pl->events[ev++];
                if (hipEventRecord(e, st[0]) != hipSuccess) return USOT_ELAUNCH;
                if (hipStreamWaitEvent(st[op.lane], e, 0) != hipSuccess) return USOT_ELAUNCH;
            }
            break;
        case K_JOIN:      // lane 0 waits for lane op.lane
            if (lanes && op.lane != 0) {
                if (ev >= pl->events.size()) return USOT_ESTATE;
                hipEvent_t e = pl->events[ev++];
                if (hipEventRecord(e, st[op.lane]) != hipSuccess) return USOT_ELAUNCH;
                if (hipStreamWaitEvent(st[0], e, 0) != hipSuccess) return USOT_ELAUNCH;
            }
            break;
        }
        if (rc != USOT_OK) return rc;
    }
    if (marks && hipEventRecord(marks[opi], main_stream) != hipSuccess) return USOT_ELAUNCH;
    return USOT_OK;
}

int prepare_lanes(Plan *pl)
{
    size_t need = 0;
    bool any = false;
    for (const Op &op : pl->ops) {
        if (op.kind == K_FORK || op.kind == K_JOIN) ++need;
        if (op.lane != 0) any = true;
    }
    if (!any) return USOT_OK;
    for (int k = 1; k < kLanes; ++k)
        if (!pl->side[k] && hipStreamCreateWithFlags(&pl->side[k], hipStreamNonBlocking) != hipSuccess)
            return USOT_ENOMEM;
    while (pl->events.size() < need) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return USOT_ENOMEM;
        pl->events.push_back(e);
    }
    return USOT_OK;
}

Op *push(void *plan, Kind k)
{
    Plan *pl = (Plan *)plan;
    if (!pl || pl->captured) return nullptr;
    Op op{};
    op.kind = k;
    op.lane = pl->cur_lane;
    pl->ops.push_back(op);
    return &pl->ops.back();
}

}  // namespace

extern "C" void *usot_plan_create(void) { return new (std::nothrow) Plan(); }

extern "C" void usot_plan_destroy(void *plan)
{
    Plan *pl = (Plan *)plan;
    if (!pl) return;
    if (pl->exec) (void)hipGraphExecDestroy(pl->exec);
    if (pl->graph) (void)hipGraphDestroy(pl->graph);
    for (hipEvent_t e : pl->events) (void)hipEventDestroy(e);
    for (int k = 1; k < kLanes; ++k)
        if (pl->side[k]) (void)hipStreamDestroy(pl->side[k]);
    delete pl;
}

extern "C" int usot_plan_size(void *plan) { return plan ? (int)((Plan *)plan)->ops.size() : USOT_EINVAL; }

extern "C" int usot_plan_add_conv(void *plan, const usot_conv_desc *d)
{
    if (!d) return USOT_EINVAL;
    return usot_plan_add_conv_batch(plan, d, 1);
}

extern "C" int usot_plan_add_conv_batch(void *plan, const usot_conv_desc *d, int n)
{
    if (!d || n < 1 || n > 4) return USOT_EINVAL;
    Op *op = push(plan, K_CONV);
    if (!op) return USOT_ESTATE;
    op->conv = d[0];
    for (int q = 1; q < n; ++q) op->more[q - 1] = d[q];
    op->nconv = n;
    return USOT_OK;
}

extern "C" int usot_plan_add_conv_lp(void *plan, const usot_conv_desc *d, int dtype, int out_f32)
{
    if (!d) return USOT_EINVAL;
    Op *op = push(plan, K_CONVB);
    if (!op) return USOT_ESTATE;
    op->conv = *d;
    op->i[6] = dtype; op->i[7] = out_f32;
    return USOT_OK;
}

extern "C" int usot_plan_add_conv_bf16(void *plan, const usot_conv_desc *d) { return usot_plan_add_conv_lp(plan, d, 0, 0); }

extern "C" int usot_plan_add_pw_pair(void *plan, const usot_pw_pair_desc *d, int dtype)
{
    if (!d || !(dtype == 3 ? usot_pw_pair_f32s_supported(d->CM, d->CO, d->CN) : dtype == 2 ? usot_pw_pair_f32_supported(d->CM, d->CO, d->CN) : usot_pw_pair_supported(d->CM, d->CO, d->CN))) return USOT_EINVAL;
    // the fp32 (256, 1024, 256) pair exists in the channel-sliced form only: it needs M small enough to slice AND a workspace —
    // refuse it here, at build time, not at the first run / graph capture
    if ((dtype == 2 || dtype == 3) && d->CM == 256 && !(d->ws && usot_pw_pair_f32_ws_floats(d->M, d->CM, d->CO, d->CN) > 0)) return USOT_EINVAL;
    Op *op = push(plan, K_PWPAIR);
    if (!op) return USOT_ESTATE;
    op->pw = *d;
    op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_pw_single(void *plan, const float *x, const float *wp, const float *b, const float *res, float *y,
                                       int M, int K, int N, int act)
{
    if (!usot_pw_single_f32_supported(K, N)) return USOT_EINVAL;
    Op *op = push(plan, K_PW1);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = wp; op->p[2] = b; op->p[3] = res; op->p[4] = y;
    op->i[0] = M; op->i[1] = K; op->i[2] = N; op->i[3] = act;
    return USOT_OK;
}

extern "C" int usot_plan_add_conv3x3_halo(void *plan, const void *x, const void *w, const float *bias, void *y,
                                          int N, int H, int W, int Cin, int Cout, int act, int dtype)
{
    if (!usot_conv3x3_halo_supported(Cin, Cout)) return USOT_EINVAL;
    Op *op = push(plan, K_HALO);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = w; op->p[2] = bias; op->p[3] = y;
    op->i[0] = N; op->i[1] = H; op->i[2] = W; op->i[3] = Cin; op->i[4] = Cout; op->i[5] = act; op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_bneck_first(void *plan, const usot_bneck_desc *d, int dtype)
{
    if (!d || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    Op *op = push(plan, K_BNECK1);
    if (!op) return USOT_ESTATE;
    op->bneck = *d;
    op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_bneck_tail(void *plan, const usot_bneck_desc *d, int Cnext, int dtype)
{
    if (!d || (dtype != 0 && dtype != 1) || !usot_bneck_tail_supported(64, 256, Cnext)) return USOT_EINVAL;
    Op *op = push(plan, K_BNECKT);
    if (!op) return USOT_ESTATE;
    op->bneck = *d;
    op->i[5] = Cnext;
    op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_conv_kstream(void *plan, const void *x, const void *w, const float *bias, void *y,
                                          int N, int H, int W, int Cin, int Cout, int stride, int pad, int dil, int act, int dtype)
{
    if (!usot_conv_kstream_supported(Cin, Cout, 3, 3)) return USOT_EINVAL;
    Op *op = push(plan, K_CKSTREAM);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = w; op->p[2] = bias; op->p[3] = y;
    op->i[0] = N; op->i[1] = H; op->i[2] = W; op->i[3] = Cin; op->i[4] = Cout; op->i[5] = stride; op->i[7] = pad; op->i[6] = dtype;
    op->l[0] = dil; op->l[1] = act;
    return USOT_OK;
}

extern "C" int usot_plan_add_pw_kstream(void *plan, const void *x, const void *w, const float *bias, void *y,
                                        long M, int K, int N, int act, int dtype)
{
    if (!usot_pw_kstream_supported(K, N)) return USOT_EINVAL;
    Op *op = push(plan, K_KSTREAM);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = w; op->p[2] = bias; op->p[3] = y;
    op->l[0] = M; op->i[1] = K; op->i[2] = N; op->i[3] = act; op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_pw_panel_pair(void *plan, const usot_pw_pair_desc *d, int dtype)
{
    if (!d || !usot_pw_panel_pair_supported(d->CM, d->CO, d->CN) || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    Op *op = push(plan, K_PANELP);
    if (!op) return USOT_ESTATE;
    op->pw = *d;
    op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_conv_pw(void *plan, const usot_conv_desc *c2, const void *w3, const float *b3, const void *res, void *y,
                                     int dtype)
{
    if (!c2 || !usot_conv_pw_supported(c2->Cin, c2->Cout, 4 * c2->Cout) || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    Op *op = push(plan, K_CONVPW);
    if (!op) return USOT_ESTATE;
    op->conv = *c2;
    op->p[0] = w3; op->p[1] = b3; op->p[2] = res; op->p[3] = y;
    op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_conv_pw_pair(void *plan, const usot_conv_desc *c2, const usot_pw_pair_desc *d, int dtype)
{
    if (!c2 || !d || !usot_conv_pw_pair_supported(d->CM, d->CO, d->CN) || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    Op *op = push(plan, K_CONVPWP);
    if (!op) return USOT_ESTATE;
    op->conv = *c2;
    op->pw = *d;
    op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_conv_pw_ov(void *plan, const usot_conv_desc *c2, const usot_pw_pair_desc *d, int dtype, void *ws)
{
    if (!c2 || !d || !ws || !usot_conv_pw_ov_supported(d->CM, d->CO, d->CN) || (dtype != 0 && dtype != 1)) return USOT_EINVAL;
    Op *op = push(plan, K_CONVPWO);
    if (!op) return USOT_ESTATE;
    op->conv = *c2;
    op->pw = *d;
    op->p[0] = ws;
    op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_pw_panel(void *plan, const void *x, const void *w, const float *bias, const void *res, void *y,
                                      int M, int K, int N, int act, int dtype)
{
    if (!usot_pw_panel_supported(K, N)) return USOT_EINVAL;
    Op *op = push(plan, K_PANEL);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = w; op->p[2] = bias; op->p[3] = res; op->p[4] = y;
    op->i[0] = M; op->i[1] = K; op->i[2] = N; op->i[3] = act; op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_pw_triple(void *plan, const float *x, const float *w2p, const float *b2, const usot_pw_pair_desc *d,
                                       int Nb, int H, int W, int Cin, int OH, int OW, int pad_h, int pad_w, int dil_h, int dil_w)
{
    if (!d || !usot_pw_triple_f32_supported(Cin, d->CM, d->CO, d->CN)) return USOT_EINVAL;
    Op *op = push(plan, K_PW3);
    if (!op) return USOT_ESTATE;
    op->pw = *d;
    op->p[0] = x; op->p[1] = w2p; op->p[2] = b2;
    op->i[0] = Nb; op->i[1] = H; op->i[2] = W; op->i[3] = Cin; op->i[4] = OH; op->i[5] = OW; op->i[6] = pad_h; op->i[7] = pad_w;
    op->l[0] = dil_h; op->l[1] = dil_w;
    return USOT_OK;
}

extern "C" int usot_plan_add_stream_conv3x3(void *plan, const float *x, const float *wp, const float *b, const float *res, float *y,
                                            int Nb, int H, int W, int Cin, int OH, int OW, int N, int pad_h, int pad_w, int dil_h,
                                            int dil_w, int act)
{
    if (!usot_stream_conv3x3_f32_supported(Cin, N)) return USOT_EINVAL;
    Op *op = push(plan, K_SC3);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = wp; op->p[2] = b; op->p[3] = res; op->p[4] = y;
    op->i[0] = Nb; op->i[1] = H; op->i[2] = W; op->i[3] = Cin; op->i[4] = OH; op->i[5] = OW; op->i[6] = N; op->i[7] = pad_h;
    op->l[0] = pad_w; op->l[1] = dil_h; op->l[2] = dil_w; op->l[3] = act;
    return USOT_OK;
}

extern "C" int usot_plan_add_cvt_lp(void *plan, const float *src, void *dst, int64_t n, int dtype)
{
    Op *op = push(plan, K_CVTB);
    if (!op) return USOT_ESTATE;
    op->p[0] = src; op->p[1] = dst; op->l[0] = n; op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_stem_pool(void *plan, const float *x, const float *wfrag, const float *bias, float *y,
                                       int N, int H, int W, int OH, int OW, int PH, int PW)
{
    return usot_plan_add_stem_pool_ind(plan, x, wfrag, bias, y, N, H, W, OH, OW, PH, PW, nullptr);
}

extern "C" int usot_plan_add_stem_pool_ind(void *plan, const float *x, const float *wfrag, const float *bias, float *y,
                                           int N, int H, int W, int OH, int OW, int PH, int PW, const int32_t *xptr_dev)
{
    return usot_plan_add_stem_pool_mu(plan, x, wfrag, bias, y, N, H, W, OH, OW, PH, PW, xptr_dev, 0.f, 0.f, 0.f);
}

extern "C" int usot_plan_add_stem_pool_mu(void *plan, const float *x, const float *wfrag, const float *bias, float *y,
                                          int N, int H, int W, int OH, int OW, int PH, int PW, const int32_t *xptr_dev,
                                          float mu0, float mu1, float mu2)
{
    Op *op = push(plan, K_STEMP);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = wfrag; op->p[2] = bias; op->p[3] = y; op->p[4] = xptr_dev;
    op->i[0] = N; op->i[1] = H; op->i[2] = W; op->i[3] = OH; op->i[4] = OW; op->i[5] = PH; op->i[6] = PW;
    op->f[1] = mu0; op->f[2] = mu1; op->f[3] = mu2;
    return USOT_OK;
}

extern "C" int usot_plan_add_thin_conv(void *plan, const usot_conv_desc *d, int n)
{
    if (!d || n < 1 || n > 4) return USOT_EINVAL;
    Op *op = push(plan, K_THIN);
    if (!op) return USOT_ESTATE;
    op->conv = d[0];
    for (int q = 1; q < n; ++q) op->more[q - 1] = d[q];
    op->nconv = n;
    return USOT_OK;
}

extern "C" int usot_plan_add_rows_copy_multi(void *plan, int nseg, const float *const *src, const int32_t *idx_dev,
                                             float *const *dst, int n_rows, const int32_t *row_len, int scatter,
                                             int32_t *stash_next)
{
    if (nseg < 1 || nseg > 4 || !src || !dst || !row_len) return USOT_EINVAL;
    Op *op = push(plan, K_ROWSM);
    if (!op) return USOT_ESTATE;
    for (int i = 0; i < nseg; ++i) { op->p[i] = src[i]; op->l[i] = (int64_t)(uintptr_t)dst[i]; op->i[3 + i] = row_len[i]; }
    op->p[4] = idx_dev; op->p[5] = stash_next;
    op->i[0] = nseg; op->i[1] = n_rows; op->i[2] = scatter;
    return USOT_OK;
}

extern "C" int usot_plan_add_rows_append_gather(void *plan, const float *const *fresh, float *const *bank, float *const *picked,
                                                const int32_t *row_len, const int32_t *idx_dev, int n_pick, int slot_pos)
{
    if (!fresh || !bank || !picked || !row_len || !idx_dev || n_pick < 1 || n_pick > 32 || slot_pos < 0) return USOT_EINVAL;
    Op *op = push(plan, K_ROWSAG);
    if (!op) return USOT_ESTATE;
    for (int i = 0; i < 4; ++i) { op->p[i] = fresh[i]; op->l[i] = (int64_t)(uintptr_t)bank[i]; op->i[2 + i] = row_len[i]; }
    for (int i = 0; i < 3; ++i) op->l[4 + i] = (int64_t)(uintptr_t)picked[i];
    op->p[4] = idx_dev;
    op->i[0] = n_pick; op->i[1] = slot_pos;
    return USOT_OK;
}

extern "C" int usot_plan_add_stem_pool_lp(void *plan, const float *x, const void *wfrag, const float *bias, void *y,
                                          int N, int H, int W, int OH, int OW, int PH, int PW, int dtype,
                                          float mu0, float mu1, float mu2)
{
    Op *op = push(plan, K_STEMB);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = wfrag; op->p[2] = bias; op->p[3] = y;
    op->i[0] = N; op->i[1] = H; op->i[2] = W; op->i[3] = OH; op->i[4] = OW; op->i[5] = PH; op->i[7] = PW; op->i[6] = dtype;
    op->f[1] = mu0; op->f[2] = mu1; op->f[3] = mu2;
    return USOT_OK;
}

extern "C" int usot_plan_add_cvt_bf16(void *plan, const float *src, void *dst, int64_t n) { return usot_plan_add_cvt_lp(plan, src, dst, n, 0); }

extern "C" int usot_plan_add_maxpool_lp(void *plan, const void *x, void *y, int N, int H, int W, int C, int OH, int OW, int dtype)
{
    Op *op = push(plan, K_POOLB);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = y;
    op->i[0] = N; op->i[1] = H; op->i[2] = W; op->i[3] = C; op->i[4] = OH; op->i[5] = OW; op->i[6] = dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_maxpool_bf16(void *plan, const void *x, void *y, int N, int H, int W, int C, int OH, int OW)
{
    return usot_plan_add_maxpool_lp(plan, x, y, N, H, W, C, OH, OW, 0);
}

extern "C" int usot_plan_add_groupdw(void *plan, const usot_groupdw_desc *d)
{
    if (!d) return USOT_EINVAL;
    return usot_plan_add_groupdw_multi(plan, d, 1);
}

extern "C" int usot_plan_add_groupdw_multi(void *plan, const usot_groupdw_desc *d, int nseg)
{
    if (!d || nseg < 1 || nseg > 3) return USOT_EINVAL;
    Op *op = push(plan, K_GDW);
    if (!op) return USOT_ESTATE;
    for (int i = 0; i < nseg; ++i) op->gdw[i] = d[i];
    op->ngdw = nseg;
    return USOT_OK;
}

extern "C" int usot_plan_add_groupdw_multi_lp(void *plan, const usot_groupdw_desc *d, int nseg, int out_dtype)
{
    if (!d || nseg < 1 || nseg > 3 || (out_dtype != 1 && out_dtype != 2)) return USOT_EINVAL;
    Op *op = push(plan, K_GDW);
    if (!op) return USOT_ESTATE;
    for (int i = 0; i < nseg; ++i) op->gdw[i] = d[i];
    op->ngdw = nseg;
    op->i[6] = out_dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_conf_reduce_lp(void *plan, const void *cv, int in_dtype, void *out, int B, int M, int P, int C, int out_dtype)
{
    if ((out_dtype != 1 && out_dtype != 2) || in_dtype < 0 || in_dtype > 2) return USOT_EINVAL;
    Op *op = push(plan, K_CONF);
    if (!op) return USOT_ESTATE;
    op->p[0] = cv; op->p[1] = out;
    op->i[0] = B; op->i[1] = M; op->i[2] = P; op->i[3] = C; op->i[5] = in_dtype; op->i[6] = out_dtype;
    return USOT_OK;
}

extern "C" int usot_plan_add_stem(void *plan, const float *x, const float *w, const float *bias, float *y,
                                  int N, int H, int W, int OH, int OW)
{
    return usot_plan_add_stem_mu(plan, x, w, bias, y, N, H, W, OH, OW, 0.f, 0.f, 0.f);
}

extern "C" int usot_plan_add_stem_mu(void *plan, const float *x, const float *w, const float *bias, float *y,
                                     int N, int H, int W, int OH, int OW, float mu0, float mu1, float mu2)
{
    Op *op = push(plan, K_STEM);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = w; op->p[2] = bias; op->p[3] = y;
    op->i[0] = N; op->i[1] = H; op->i[2] = W; op->i[3] = OH; op->i[4] = OW;
    op->f[1] = mu0; op->f[2] = mu1; op->f[3] = mu2;
    return USOT_OK;
}

extern "C" int usot_plan_add_maxpool(void *plan, const float *x, float *y, int N, int H, int W, int C,
                                     int OH, int OW)
{
    Op *op = push(plan, K_POOL);
    if (!op) return USOT_ESTATE;
    op->p[0] = x; op->p[1] = y;
    op->i[0] = N; op->i[1] = H; op->i[2] = W; op->i[3] = C; op->i[4] = OH; op->i[5] = OW;
    return USOT_OK;
}

extern "C" int usot_plan_add_conf_reduce(void *plan, const float *cv, float *out, int B, int M, int P, int C)
{
    Op *op = push(plan, K_CONF);
    if (!op) return USOT_ESTATE;
    op->p[0] = cv; op->p[1] = out;
    op->i[0] = B; op->i[1] = M; op->i[2] = P; op->i[3] = C;
    return USOT_OK;
}

extern "C" int usot_plan_add_prroi(void *plan, const float *feat, const float *rois, float *out,
                                   int R, int C, int H, int W, int PH, int PW, float scale,
                                   int64_t f_sb, int64_t f_sc, int64_t f_sh, int64_t f_sw,
                                   int64_t o_sr, int64_t o_sc, int64_t o_sh, int64_t o_sw)
{
    Op *op = push(plan, K_PRROI);
    if (!op) return USOT_ESTATE;
    op->p[0] = feat; op->p[1] = rois; op->p[2] = out;
    op->i[0] = R; op->i[1] = C; op->i[2] = H; op->i[3] = W; op->i[4] = PH; op->i[5] = PW;
    op->f[0] = scale;
    op->l[0] = f_sb; op->l[1] = f_sc; op->l[2] = f_sh; op->l[3] = f_sw;
    op->l[4] = o_sr; op->l[5] = o_sc; op->l[6] = o_sh; op->l[7] = o_sw;
    return USOT_OK;
}

extern "C" int usot_plan_add_permute(void *plan, const float *src, float *dst, int D0, int D1, int D2, int D3,
                                     int64_t s0, int64_t s1, int64_t s2, int64_t s3)
{
    Op *op = push(plan, K_PERM);
    if (!op) return USOT_ESTATE;
    op->p[0] = src; op->p[1] = dst;
    op->i[0] = D0; op->i[1] = D1; op->i[2] = D2; op->i[3] = D3;
    op->l[0] = s0; op->l[1] = s1; op->l[2] = s2; op->l[3] = s3;
    return USOT_OK;
}

/* tsz_dev: device double[2] = target size * scale_z, refreshed by the host per frame;
 * roi_out: optional device float[5], receives (0, pool_label_search(best box)).          */
extern "C" int usot_plan_add_decode(void *plan, const float *cls, const float *cls_mem, const float *bbox,
                                    const double *window, double *out, int S, int instance_size, int stride,
                                    float ratio, double penalty_k, double window_influence,
                                    const double *tsz_dev, float *roi_out)
{
    Op *op = push(plan, K_DECODE);
    if (!op) return USOT_ESTATE;
    op->p[0] = cls; op->p[1] = cls_mem; op->p[2] = bbox; op->p[3] = window; op->p[4] = out; op->p[5] = tsz_dev;
    op->i[0] = S; op->i[1] = instance_size; op->i[2] = stride;
    op->f[0] = ratio; op->d[0] = penalty_k; op->d[1] = window_influence;
    op->l[0] = (int64_t)(uintptr_t)roi_out;
    return USOT_OK;
}

extern "C" int usot_plan_add_rows_copy(void *plan, const float *src, const int32_t *idx_dev, float *dst,
                                       int n_rows, int row_len, int scatter)
{
    Op *op = push(plan, K_ROWS);
    if (!op) return USOT_ESTATE;
    op->p[0] = src; op->p[1] = idx_dev; op->p[2] = dst;
    op->i[0] = n_rows; op->i[1] = row_len; op->i[2] = scatter;
    return USOT_OK;
}

extern "C" int usot_plan_fork(void *plan, int lane)
{
    Plan *pl = (Plan *)plan;
    if (!pl || lane < 0 || lane >= kLanes) return USOT_EINVAL;
    pl->cur_lane = lane;
    Op *op = push(plan, K_FORK);
    if (!op) return USOT_ESTATE;
    return USOT_OK;
}

extern "C" int usot_plan_join(void *plan, int lane)
{
    Plan *pl = (Plan *)plan;
    if (!pl || lane < 0 || lane >= kLanes) return USOT_EINVAL;
    pl->cur_lane = lane;
    Op *op = push(plan, K_JOIN);
    if (!op) return USOT_ESTATE;
    pl->cur_lane = 0;
    return USOT_OK;
}

extern "C" int usot_plan_capture(void *plan, void *stream)
{
    Plan *pl = (Plan *)plan;
    if (!pl || pl->captured || pl->ops.empty()) return USOT_ESTATE;
    int rc = prepare_lanes(pl);
    if (rc != USOT_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return USOT_ELAUNCH;
    rc = issue(pl, s, true);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc != USOT_OK || e != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return rc != USOT_OK ? rc : USOT_ELAUNCH;
    }
    if (hipGraphInstantiate(&pl->exec, g, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(g);
        return USOT_ELAUNCH;
    }
    pl->graph = g;
    pl->captured = true;
    return USOT_OK;
}

extern "C" int usot_plan_run(void *plan, void *stream)
{
    Plan *pl = (Plan *)plan;
    if (!pl) return USOT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (pl->captured) return hipGraphLaunch(pl->exec, s) == hipSuccess ? USOT_OK : USOT_ELAUNCH;
    return issue(pl, s, false);
}

/* Per-op timing with HIP events on the launching stream: issues the plan `frames` times in
 * program order on `stream` (lanes ignored, no graph) with an event before every op, each op
 * launched `reps` times back to back (amortises the event's own cost), and returns the mean
 * milliseconds per launch in ms_per_op[usot_plan_size()].
 * Blocks until done.  Used by bench.py for the roofline object.                           */
extern "C" int usot_plan_profile(void *plan, void *stream, int frames, int reps, float *ms_per_op)
{
    Plan *pl = (Plan *)plan;
    if (!pl || !ms_per_op || frames < 1 || reps < 1 || pl->ops.empty()) return USOT_EINVAL;
    const size_t n = pl->ops.size();
    std::vector<hipEvent_t> marks(n + 1);
    for (auto &e : marks)
        if (hipEventCreate(&e) != hipSuccess) return USOT_ENOMEM;
    std::vector<double> acc(n, 0.0);
    hipStream_t s = (hipStream_t)stream;
    int rc = USOT_OK;
    for (int f = 0; f < frames && rc == USOT_OK; ++f) {
        rc = issue(pl, s, false, marks.data(), reps);
        if (rc != USOT_OK) break;
        if (hipStreamSynchronize(s) != hipSuccess) { rc = USOT_ELAUNCH; break; }
        for (size_t i = 0; i < n; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, marks[i], marks[i + 1]) != hipSuccess) { rc = USOT_ELAUNCH; break; }
            acc[i] += ms;
        }
    }
    for (auto &e : marks) (void)hipEventDestroy(e);
    if (rc == USOT_OK)
        for (size_t i = 0; i < n; ++i) ms_per_op[i] = (float)(acc[i] / frames / reps);
    return rc;
}

/* kind of op i: 0 conv, 1 stem, 2 maxpool, 3 groupdw, 4 conf_reduce, 5 prroi, 6 permute,
 * 7 decode, 8 fork, 9 join; for convs also the tile the launcher would pick (info[0..3] =
 * kind, tile, ksplit, groups). */
extern "C" int usot_plan_op_info(void *plan, int i, int *info)
{
    Plan *pl = (Plan *)plan;
    if (!pl || !info || i < 0 || i >= (int)pl->ops.size()) return USOT_EINVAL;
    const Op &op = pl->ops[i];
    info[0] = (int)op.kind; info[1] = 0; info[2] = 1; info[3] = 1;
    if (op.kind == K_CONVB) { info[1] = op.conv.tile; }
    if (op.kind == K_CONV) {
        info[1] = usot_conv_resolve_tile(&op.conv);
        info[2] = op.conv.ksplit > 1 ? op.conv.ksplit : 1;
        info[3] = op.conv.groups;
    }
    return USOT_OK;
}
