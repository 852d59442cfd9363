#!/usr/bin/env python3
"""Headline benchmark: tracker FPS (255x255 search, ResNet-50) on N MI355X.

A "step" is one tracked frame per rank: gather the N_q = 7 memory kernels, ResNet-50
(layer3) backbone + neck over a resident 255x255 fp32 crop, cls/reg/memory heads with the
fused depthwise xcorr, on-device decode, PrRoIPool of the new memory feature — the whole
per-frame device path of `USOTTracker.track`, replayed as one hipGraph, with the 64-byte
result read back to the host (the tracking loop is sequential per stream, so that sync is
part of a frame).  Workload = BASELINE.json configs[1] (batch 1, fp32, one stream per GPU).
Weights are synthetic (no checkpoint ships with the reference), broadcast once from rank 0
over RCCL; streams are independent, so N GPUs = N streams, no data-path collective
("scaling": "weak").

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the fp32-MFMA implicit-GEMM
convolution, per-launch time from HIP events on the launching stream) and `cpu_baseline`
(the CPU oracle restatement of the same frame, timed on this box's host cores).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from usot_amd import hip, streams, synth  # noqa: E402
from usot_amd.model import USOT  # noqa: E402
from usot_amd.tracker import USOTConfig, select_memory  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD
LP_PEAK_TFLOPS = 2500.0          # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF is 2:1 sparse)
HBM_PEAK_GBPS = 8000.0
GROUPDW_BYTES_PER_SAMPLE = 3161088   # x 2 464 768 + k 56 320 + out 640 000


def build_model(rank, world, device):
    m = USOT()
    if rank == 0:
        m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True), strict=True)
    m.eval()
    m = m.to(device)
    nbytes = streams.broadcast_weights(m, src=0, device=device)
    return m, nbytes


def f32_dtype_label(engine):
    """`dtype` of an fp32-storage workload, from the engine options it actually ran with: 'f32' = exact fp32 products
    (v_mfma_f32_16x16x4_f32, the reference's arithmetic); the opt-in split-fp16 mode says what it is."""
    if engine.opt.get('split16_f32'):
        return ('f32 storage / accumulation, split-fp16 products (operands as hi + lo fp16: 22 significant bits, |x| < 8 188 with automatic '
                'exact-fp32 fallback) in the K >= %d convs and layer3\'s fused 1x1 pairs' % engine.opt.get('split16_min_k', 0))
    return 'f32'


def open_stream(model, device, seed, size=255):
    """Template + memory seeds for one synthetic video stream; returns (session, crops)."""
    p = USOTConfig()
    p.instance_size = size
    p.renew()
    p.sf_size = p.score_size
    t = lambda a: torch.from_numpy(a).to(device)
    model.pr_pool = True
    model.template(t(synth.crop(1000 + seed, 1, 127)), template_bbox=torch.tensor([[3.5, 3.5, 10.5, 10.5]], device=device))
    crops = t(synth.crop(2000 + seed, 8, size))
    roi = torch.tensor([[9.0, 9.0, 16.0, 16.0]], device=device)
    feats = [model.extract_memory_feature(ori_x=crops[0:1], search_bbox=roi),
             model.extract_memory_feature(ori_x=crops[0:1].flip(3), search_bbox=roi)]
    window = np.outer(np.hanning(p.score_size), np.hanning(p.score_size))
    model.engine.session_capacity = 16384     # bank rows: the timed region must not regrow (and re-capture) the session
    return model.engine.open_session(p, window, feats), crops, p


class Confidences:
    """state['memory_confidences'] of the benchmark loop: the tracker's numpy mirror of the list
    (USOTTracker._conf_array) without the list — append + a view of the filled part."""

    def __init__(self, first=0.9, cap=1 << 16):
        self.buf = np.empty(cap, np.float64)
        self.buf[0], self.n = first, 1

    def append(self, v):
        if self.n == len(self.buf):
            self.buf = np.concatenate([self.buf, np.empty_like(self.buf)])
        self.buf[self.n] = v
        self.n += 1

    def view(self):
        return self.buf[:self.n]


def run_frames(sess, crops, p, conf, n):
    for i in range(n):
        picks = select_memory(conf.view(), p.mem_queue_size)
        out = sess.frame(crops[i % crops.shape[0]], picks, (63.5, 63.5), inplace=True)     # resident crops, read where they lie
        conf.append(float(out[1]))


def run_frames_multi(group, n):
    """`group`: [(session, crops, p, conf, stream)] — S independent videos on one GPU, each on its
    own HIP stream: submit a frame for every stream, then collect them (frames of different
    videos overlap on the GPU; each video stays sequential)."""
    for i in range(n):
        for sess, crops, p, conf, st in group:
            with torch.cuda.stream(st):
                sess.submit(crops[i % crops.shape[0]], select_memory(conf.view(), p.mem_queue_size), (63.5, 63.5), inplace=True)
        for sess, crops, p, conf, st in group:
            conf.append(float(sess.collect()[1]))


K_PWPAIR, K_PW1, K_SC3, K_PW3 = 18, 19, 20, 21          # plan op kinds of the fused pointwise pair / the streaming 1x1 and 3x3 convs (csrc/plan.hip)


def build_info_line():
    """What the in-tree library is and whether the last build() compiled anything (usot_amd/build.py: build_info.json)."""
    from usot_amd import build
    bi = build.build_info()
    return {'lib': os.path.relpath(build.LIB, ROOT), 'lib_mtime': bi.get('lib_mtime'), 'lib_stale': bi.get('lib_stale'),
            'csrc_tree': bi.get('csrc_tree_now'), 'built_from_csrc_tree': bi.get('csrc_tree'),
            'compiled_units_last_build': bi.get('compiled_units'), 'build_seconds': bi.get('seconds'), 'build_host': bi.get('host'),
            'build_when': bi.get('when')}


def _tree_now():
    from usot_amd import build
    if not hasattr(_tree_now, 'v'):
        _tree_now.v = build.csrc_tree()
    return _tree_now.v


def counters_current(meta, path):
    """(ok, source string).  A committed counter file (profiles/pmc_*.json) describes the kernels of ONE source tree: its
    `_meta.csrc_tree` (usot_amd/build.py: csrc_tree, a content hash of csrc/ + include/) must equal the running tree's, otherwise
    every field derived from it is reported as null with the reason — a kernel edit without a new scripts/round_snapshot.sh
    pass can no longer leave stale traffic / mfma_busy / clock figures in the line."""
    have, now = (meta or {}).get('csrc_tree'), _tree_now()
    if have == now:
        return True, '%s@%s (csrc_tree %s)' % (path, (meta or {}).get('commit', ''), now)
    return False, '%s is STALE: measured on csrc_tree %s, running %s - rerun scripts/round_snapshot.sh' % (path, have, now)


def roofline(sess, frames):
    """Dominant kernel = the conv_igemm_f32 tile instance with the largest total time."""
    prof = sess.plan.profile(frames=frames, reps=1)      # every op once per pass, in frame order (cold operands, as in a replay)
    tiles = hip.tile_table()
    convs = iter(zip(sess.log, sess.f32_bytes))
    agg, total_ms, conv_ms, conv_flops, conv_bytes = {}, 0.0, 0.0, 0.0, 0
    for kind, tile, ks, groups, ms in prof:
        total_ms += ms
        if kind not in (0, K_PWPAIR, K_PW1, K_SC3, K_PW3):
            continue
        (name, M, N, K, g, macs), nbytes = next(convs)
        conv_ms += ms
        conv_flops += 2.0 * macs
        conv_bytes += nbytes
        if kind in (K_PWPAIR, K_PW1, K_SC3, K_PW3):   # csrc/smallm_f32.hip kernels: counted in all_convs,
            continue                              # not a tile instance of the conv_igemm family
        a = agg.setdefault(tile, [0, 0.0, 0.0, 0])
        a[0] += 1
        a[1] += ms
        a[2] += 2.0 * macs
        a[3] += nbytes
    tile, (n, ms, fl, alg_bytes) = max(agg.items(), key=lambda kv: kv[1][1])
    ach = fl / (ms * 1e-3) / 1e12
    # HBM bytes per launch cannot be counted from inside this process: they come from the committed
    # rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by scripts/pmc_to_traffic.py with the
    # source commit recorded); a kernel that was not profiled there reports null, never a stale number.
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            pmc = json.load(f)
        ok, traffic_src = counters_current(pmc.get('_meta'), 'profiles/pmc_traffic.json')
        if ok:
            traffic = pmc.get(hip.tile_name(tile), {}).get('hbm_bytes_per_launch')
            traffic_src += ' (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)'
    except Exception:
        pass
    # the peak that bounds the dominant kernel's ARITHMETIC: the fp32 MFMA (157.3 TFLOP/s) for the exact-fp32 tiles; for the
    # split-fp16 tiles (hip.tile_wfrag == 2: every product block as three v_mfma_f32_16x16x32_f16, csrc/conv_igemm.hip PF = 4) the
    # dense fp16 peak over the three products an algorithmic product costs - `achieved` stays ALGORITHMIC flops per second
    split16 = hip.tile_wfrag(tile) == 2
    peak = round(LP_PEAK_TFLOPS / 3.0, 1) if split16 else MFMA_F32_PEAK_TFLOPS
    busy = pmc_busy(hip.tile_name(tile), peak)
    return {
        'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
        'frac': round(ach / peak, 4), **busy_fields(busy, ach),
        'arithmetic': ('split fp16: operands as hi + lo fp16 (22 significant bits), 3 x v_mfma_f32_16x16x32_f16 per product block, fp32 '
                       'accumulation; peak = dense fp16 MFMA peak %.0f / 3' % LP_PEAK_TFLOPS) if split16 else 'v_mfma_f32_16x16x4_f32 (exact fp32)',
        'executed_tflops': round(ach * (3 if split16 else 1), 2), 'frac_of_fp32_mfma_peak': round(ach / MFMA_F32_PEAK_TFLOPS, 4),
        'traffic': traffic, 'traffic_source': traffic_src,
        # algorithmic bytes: every launch's input map, filter bank, bias and result (+ residual) once — SURVEY 8(d); the
        # measured traffic above it is Infinity-Cache-served re-reads of the activations by each XCD's L2 (~150 FLOP/B:
        # not the limiter of this kernel)
        'algorithmic_bytes_per_launch': int(alg_bytes / n),
        'traffic_to_algorithmic': round(traffic / (alg_bytes / n), 3) if traffic else None,
        'kernel': hip.tile_name(tile), 'launches_per_frame': n,
        'avg_launch_us': round(ms / n * 1e3, 2), 'algorithmic_gflop_per_frame': round(fl / 1e9, 3),
        'all_convs': {'gflop_per_frame': round(conv_flops / 1e9, 3), 'us_per_frame': round(conv_ms * 1e3, 1),
                      'tflops': round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                      'frac': round(conv_flops / (conv_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                      'frac_is': 'of the fp32 MFMA peak %.1f (the launches mix exact-fp32 and split-fp16 tiles)' % MFMA_F32_PEAK_TFLOPS},
        'frame_op_spans_us': round(total_ms * 1e3, 1),
    }


def pmc_busy(kernel, peak):
    """Clock and matrix-pipe utilisation of `kernel` from the committed `rocprofv3 --pmc SQ_BUSY_CYCLES
    SQ_VALU_MFMA_BUSY_CYCLES` pass (profiles/pmc_busy.json, scripts/pmc_busy.py): the sustained clock under this kernel's
    load is SQ_BUSY_CYCLES / duration, and `peak_sustained` = the nominal peak scaled by it (the nominal peak assumes
    2.4 GHz; MFMA-dense kernels run at 1.8-2.1 GHz).  None when that kernel was not profiled."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_busy.json')) as f:
            pj = json.load(f)
        ok, src = counters_current(pj.get('_meta'), 'profiles/pmc_busy.json')
        if not ok:
            return {'stale': src}
        e = pj['kernels'][kernel]
        return {'clock_ghz': e['clock_ghz'], 'mfma_busy': e['mfma_busy_frac'], 'peak_sustained': round(peak * e['clock_ghz'] / 2.4, 1),
                'source': src}
    except Exception:
        return None


def busy_fields(busy, ach):
    if not busy:
        return {'peak_sustained': None}
    if 'stale' in busy:
        return {'peak_sustained': None, 'clock_ghz': None, 'mfma_busy': None, 'busy_source': busy['stale']}
    return {'peak_sustained': busy['peak_sustained'], 'frac_of_sustained': round(ach / busy['peak_sustained'], 4),
            'clock_ghz': busy['clock_ghz'], 'mfma_busy': busy['mfma_busy'], 'busy_source': busy['source']}


def xcorr_bandwidth(device, sizes=(2048, 128), iters=20):
    """Fused GroupDW far beyond the 256 MiB Infinity Cache: achieved GB/s on the ALGORITHMIC bytes of
    SURVEY §8(d) (3 161 088 B per sample), HIP events on the launching stream.  `achieved` is the
    steady-state figure (2048 samples = 6.5 GB moved per launch); `by_samples` adds the 405 MB case
    (128 samples = exactly one wave of workgroups, so ramp-up and drain are not amortised)."""
    geo = ((5, 5), (3, 5), (5, 3))
    w = np.array([0.3, 0.3, 0.4], np.float32)
    rows = []
    for samples in sizes:
        g = torch.Generator(device=device).manual_seed(7)
        xs = [torch.randn(samples, 25 + hk - 1, 25 + wk - 1, 256, generator=g, device=device) for hk, wk in geo]
        zs = [torch.randn(samples, hk, wk, 256, generator=g, device=device) for hk, wk in geo]
        for _ in range(3):
            hip.groupdw(xs, zs, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            hip.groupdw(xs, zs, w)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        rows.append({'samples': samples, 'kernel': hip.groupdw_variant_name(samples), 'ms': round(ms, 4),
                     'bytes': samples * GROUPDW_BYTES_PER_SAMPLE,
                     'achieved': round(samples * GROUPDW_BYTES_PER_SAMPLE / (ms * 1e-3) / 1e9, 1)})
        del xs, zs
    top = rows[0]
    probe = hbm_ceiling_probe(device)
    return {'bound': 'hbm', 'kernel': top['kernel'], 'samples': top['samples'], 'ms': top['ms'],
            'achieved': top['achieved'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(top['achieved'] / HBM_PEAK_GBPS, 4),
            'ceiling_probe': probe, 'frac_of_measured_ceiling': round(top['achieved'] / probe['mix_4r_1w'], 4),
            'frac_of_pattern_emulation': round(top['achieved'] / probe['groupdw_pattern'], 4),
            'algorithmic_bytes': top['bytes'], **xcorr_traffic(top['kernel'], top['samples']), 'by_samples': rows}


def hbm_ceiling_probe(device, gib=2, iters=10):
    """What THIS box's HBM delivers (csrc/bw_probe.hip, 16 bytes per lane, far beyond the 256 MiB Infinity Cache): read-only,
    copy, and GroupDW's byte mix — 4 bytes read per byte written, one interleaved read stream, non-temporal stores.  GB/s of
    bytes moved (read + written)."""
    n = gib << 30
    src = torch.empty(n // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty(n // 4, dtype=torch.float32, device=device)
    out = {}
    for name, mode, moved in (('read', 0, n), ('copy', 1, 2 * n), ('mix_4r_1w', 2, n + n // 4)):
        run = lambda: hip.check(hip.lib().usot_bw_probe(hip.stream(), hip.ptr(src), hip.ptr(dst), n, mode), 'usot_bw_probe')
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(moved / (e0.elapsed_time(e1) / iters * 1e-3) / 1e9, 1)
    # GroupDW's own traffic pattern without its compute (mode 3): what (sample, 64-channel group) blocks streaming 256-byte
    # granules of three maps + one output map reach on this box - the kernel's structural ceiling (scripts/probes/granule_probe.hip)
    S = 2048                                                 # the sample count `xcorr_hbm.achieved` is measured at
    del src, dst
    src = torch.empty(3 * S * 841 * 256, dtype=torch.float32, device=device).fill_(1.0)
    dst = torch.empty(S * 625 * 256, dtype=torch.float32, device=device)
    run = lambda: hip.check(hip.lib().usot_bw_probe(hip.stream(), hip.ptr(src), hip.ptr(dst), S * 3 * 841 * 1024, 3), 'usot_bw_probe')
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    out['groupdw_pattern'] = round(S * (3 * 841 + 625) * 1024 / (e0.elapsed_time(e1) / iters * 1e-3) / 1e9, 1)
    out['unit'] = 'GB/s'
    out['what'] = ('%d GiB buffers, 16 B per lane; read = 8 non-temporal loads in flight per lane; copy = 32 KiB block spans, non-temporal '
                   'loads and stores; mix_4r_1w = GroupDW byte mix (one interleaved read stream, non-temporal stores); groupdw_pattern = '
                   'an address-level emulation of the GroupDW launch itself (%d samples: its blocks, its 256-byte granules, its 4 : 1 mix, '
                   'no compute)' % (gib, S))
    return out


def xcorr_traffic(kernel, samples):
    """HBM bytes per launch of the GroupDW kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_xcorr.json: FETCH_SIZE x 2 + WRITE_SIZE per sample, scripts/pmc_xcorr_to_json.py); null
    when that kernel was not profiled."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_xcorr.json')) as f:
            pmc = json.load(f)
        ok, src = counters_current(pmc.get('_meta'), 'profiles/pmc_xcorr.json')
        if not ok:
            return {'traffic': None, 'traffic_source': src}
        per = pmc[kernel]['hbm_bytes_per_sample']
        return {'traffic': int(per) * samples, 'traffic_to_algorithmic': pmc[kernel]['ratio_to_algorithmic'], 'traffic_source': src}
    except Exception:
        return {'traffic': None}


def video_loop(model, device, frames=200):
    """PCIe-inclusive rate of the real API: USOTTracker.track on host uint8 frames (upload,
    device crop, frame graph, result poll, host state update).  Not `value`."""
    from usot_amd.tracker import USOTTracker

    class Info:
        arch = 'USOT'
    trk = USOTTracker(Info())
    ims = [np.ascontiguousarray(synth.frame(77, t=t)[0]) for t in range(16)]      # dense HWC uint8, as cv2.imread returns
    im0, (cx, cy) = synth.frame(77, t=0)
    state = trk.init(im0, np.array([cx, cy]), np.array([52.0, 38.0]), model)
    for i in range(10):
        state = trk.track(state, ims[i % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        state = trk.track(state, ims[i % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'value': round(frames / dt, 1), 'unit': 'frames/s', 'frames': frames, 'frame_hw': list(ims[0].shape[:2]),
            'what': 'USOTTracker.track(state, uint8 frame): H2D upload + device crop + frame graph + poll + host update'}


def host_threads(cap=32):
    """Threads for the CPU leg: the cores this process may actually use (affinity mask and
    cgroup quota), capped — torch-CPU on every hardware thread of a 2-socket host thrashes."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def cpu_model():
    """CPU model string of the host (SURVEY §8d asks for it beside the core count)."""
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.lower().startswith('model name'):
                    return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or 'unknown'


def cpu_baseline(budget_s=14.0):
    """The oracle's restatement of one tracked frame (models.py:179-198 + PrPool) on the host, at k = all the cores this
    process may use (the headline `value` / `cores`) and at k = 1 (`single_thread`): SURVEY §8(d)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import usot_oracle as orc
    m = USOT()
    sd = synth.torch_state_dict(m, seed=0, calibrated=True)
    t = torch.from_numpy
    z, x = t(synth.crop(1000, 1, 127)), t(synth.crop(2000, 1, 255))

    def timed(threads, budget, cap):
        torch.set_num_threads(threads)
        with torch.no_grad():
            zf = orc.template(sd, z, torch.tensor([[3.5, 3.5, 10.5, 10.5]]), pr_pool=True)
            mem = torch.cat([zf] * 7, 0)
            frame = lambda: orc.prpool_feature(orc.track(sd, x, zf, mem, torch.ones(1, 7))[3], torch.tensor([[9.0, 9.0, 16.0, 16.0]]))
            t0 = time.perf_counter()
            frame()                                   # warm-up (also guards the time budget)
            if time.perf_counter() - t0 < budget / 4:
                frame()
            n, t0 = 0, time.perf_counter()
            while True:
                frame()
                n += 1
                dt = time.perf_counter() - t0
                if dt > budget or n >= cap:
                    break
        return n, dt, torch.get_num_threads()
    cores = host_threads()
    n, dt, used = timed(cores, budget_s * 0.55, 200)
    n1, dt1, _ = timed(1, budget_s * 0.45, 12)
    torch.set_num_threads(cores)
    what = ('the torch-CPU oracle restatement under no_grad with the duplicate search-side encodes of connect.py:251-264 '
            'computed once (the reference runs them three times), i.e. a faster CPU path than the literal reference')
    return {'value': round(n / dt, 3), 'unit': 'frames/s', 'cores': used, 'kind': 'port', 'cpu_model': cpu_model(),
            'sample': '%d frames of the same workload (1 crop 255x255, N_q=7, fp32) in %.1f s: %s' % (n, dt, what),
            'single_thread': {'value': round(n1 / dt1, 3), 'unit': 'frames/s', 'cores': 1,
                              'sample': '%d frames in %.1f s, torch.set_num_threads(1)' % (n1, dt1)}}


LP_DOMINANT_KERNEL = 'conv_igemm_bf16<256, 256, 4, 4, false, 0, 2, 16, 1>'     # as scripts/pmc_busy.py keys it
LP_FUSED_KERNEL = 'conv_pw_kernel<false, 16, 256, 256, true>'    # a fused layer3 block: the kernel with the most time in the step
BF16_PEAK_TFLOPS = 2500.0         # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF is 2:1 sparse)
BACKBONE_GFLOP = 28.192642        # one 255^2 crop through stem..layer3


def _timed(run, min_seconds, steps=0):
    """Run `run()` at least `steps` times and for at least `min_seconds`; returns (n, seconds)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    est = max(time.perf_counter() - t0, 1e-6)
    n = max(int(steps), int(math.ceil(min_seconds / est)), 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return n, time.perf_counter() - t0


def measure_backbone_bf16(model, device, batch=64, size=255, steps=0, warmup=2, min_seconds=1.0, top=5):
    """BASELINE configs[2]: batch-64 bf16 backbone (+neck) on v_mfma_f32_16x16x32_bf16, the MFMA-roofline
    run.  Whole-graph replay for crops/s; per-conv HIP-event times (plan.profile, each op once per
    pass in graph order) for TFLOP/s over the convolutions."""
    e = model.engine
    x = torch.from_numpy(synth.crop(3000, batch, size)).to(device)
    for _ in range(max(2, warmup)):
        e.features_bf16(x)
    p = next(v for k, v in e._feat.items() if k[:3] == ('bf16', batch, size))
    n, dt = _timed(p['plan'].run, min_seconds, steps)
    prof = stable_profile(p['plan'], dt / n * 1e3)
    convs = iter(p['log'])
    ms_conv = fl_conv = ms_all = 0.0
    rows = []
    for kind, tile, ks, groups, ms in prof:
        ms_all += ms
        if kind in (11, 18, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31):                             # K_CONVB, K_PWPAIR, K_PANEL
            name, M, N, K, g, macs = next(convs)
            ms_conv += ms
            fl_conv += 2.0 * macs
            rows.append((ms, name, M, N, K, 2.0 * macs / ms / 1e9))
    # `achieved` = SURVEY 8(d)'s algorithmic work of the step (batch x 28.192642 GFLOP: stem .. layer3, the neck's 0.504 GFLOP per crop
    # not counted) over the STEP time of the graph replay - not over the sum of the conv launches' spans, which is reported beside it
    ach = batch * BACKBONE_GFLOP * n / dt / 1e3
    conv_tflops = fl_conv / (ms_conv * 1e-3) / 1e12
    # algorithmic HBM bytes of the step: every conv launch's operands and result once (engine.Builder.lp_bytes) + the stem's
    # fp32 crops in and pooled map out; measured traffic from the committed PMC passes (scripts/pmc_lp_traffic.py)
    ph = ((size - 7) // 2 + 1 - 1) // 2 + 1
    alg_bytes = sum(p.get('lp_bytes', [])) + batch * (3 * size * size * 4 + ph * ph * 64 * 2)
    traffic, tsrc, by_kernel = None, None, None
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic_bf16.json')
    if batch == 64 and size == 255 and os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        ok, tsrc = counters_current(tj.get('_meta'), 'profiles/pmc_traffic_bf16.json')
        if ok:
            traffic = tj.get('hbm_bytes_per_step')
            tsrc += ' (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, summed over the step)'
            by_kernel = {k: v['hbm_bytes_per_launch'] for k, v in sorted(tj.get('by_kernel', {}).items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches_per_step'])[:6]}
    return {
        'workload': 'configs[2]: batch=%d search crops %dx%d bf16, backbone + neck convs on v_mfma_f32_16x16x32_bf16, '
                    'fp32 accumulate, one hipGraph' % (batch, size, size),
        'value': round(batch * n / dt, 1), 'unit': 'crops/s', 'steps': n, 'ms_per_step': round(dt / n * 1e3, 3),
        'roofline': {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(ach / BF16_PEAK_TFLOPS, 4),
                     # clock / matrix-pipe utilisation of the step's dominant MFMA kernel (the 256 x 256 tile of the K >= 2304 convs)
                     **busy_fields(pmc_busy(LP_DOMINANT_KERNEL, BF16_PEAK_TFLOPS), ach),
                     'traffic': traffic, 'traffic_source': tsrc,
                     'algorithmic_bytes_per_step': int(alg_bytes),
                     'traffic_to_algorithmic': round(traffic / alg_bytes, 3) if traffic else None,
                     # the fused layer3 blocks read their own Y panel back for the next conv1 (phase 5 of csrc/conv_pw_lp.hip): real
                     # HBM / Infinity-Cache traffic that the unfused decomposition counted as conv1's algorithmic input
                     # the fused layer3 block (conv2 -> conv3 -> next conv1, six launches per step): the kernel with the largest share of
                     # the step; its matrix pipe idles through the HBM phases (phase 4: residual + Y, phase 5: the Y readback)
                     'fused_block': (lambda b: None if not b else {'kernel': LP_FUSED_KERNEL, **{k: b.get(k) for k in ('clock_ghz', 'mfma_busy', 'stale')
                                                                                                   if k in b}})(pmc_busy(LP_FUSED_KERNEL, BF16_PEAK_TFLOPS)),
                     'fused_readback_bytes_per_step': int(sum(p.get('lp_readback', []))),
                     'traffic_to_algorithmic_plus_readback': round(traffic / (alg_bytes + sum(p.get('lp_readback', []))), 3) if traffic else None,
                     'hbm_gbs_at_algorithmic_bytes': round(alg_bytes / (dt / n) / 1e9, 1),
                     'traffic_per_launch_top_kernels': by_kernel,
                     'kernel': 'conv_igemm_bf16 / conv_pw / bneck family (all %d conv launches; a fused bottleneck launch counts once)' % len(rows),
                     'algorithmic_gflop_per_step': round(batch * BACKBONE_GFLOP, 1),
                     'conv_launches': {'gflop_per_step': round(fl_conv / 1e9, 1), 'ms_per_step': round(ms_conv, 3), 'tflops': round(conv_tflops, 1),
                                       'what': 'every conv launch behind the stem incl. the neck (HIP-event spans, plan.profile); the stem is its own launch'},
                     'conv_ms_per_step': round(ms_conv, 3),
                     'all_ops_ms_per_step': round(ms_all, 3),
                     'end_to_end_tflops': round(batch * (BACKBONE_GFLOP + 0.504) * n / dt / 1e3, 1),
                     'slowest': [{'op': nm, 'M': M, 'N': N, 'K': K, 'ms': round(ms, 3), 'tflops': round(tf, 1)}
                                 for ms, nm, M, N, K, tf in sorted(rows, reverse=True)[:top]]},
    }



def stable_profile(plan, step_ms, frames=3, tries=4):
    """Per-op times of a plan (eager, HIP events).  A host hiccup during one of the few eager frames lands in one op's mean (seen: a
    77 us launch reported as 324 us): keep the pass whose op times sum closest to - and accept the first within 6 % of - the
    graph-timed step."""
    best = None
    for _ in range(tries):
        prof = plan.profile(frames=frames, reps=1)
        tot = sum(ms for *_, ms in prof)
        if best is None or abs(tot - step_ms) < abs(best[0] - step_ms):
            best = (tot, prof)
        if abs(tot - step_ms) <= 0.06 * step_ms:
            break
    return best[1]

def backbone_bf16(a, device):
    """`--workload backbone_bf16`: configs[2] as its own JSON line."""
    model, _ = build_model(0, 1, device)
    r = measure_backbone_bf16(model, device, a.batch, a.size, a.steps, a.warmup, a.min_seconds)
    line = {
        'metric': 'backbone crops/s (255x255, ResNet-50 layer3 + neck, bf16, batch %d)' % a.batch,
        'value': r['value'], 'unit': 'crops/s', 'n_gpus': 1, 'steps': r['steps'], 'warmup': a.warmup,
        'ms_per_step': r['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': r['workload'], 'search': a.size, 'hipgraph': True},
        'roofline': r['roofline'], 'build_info': build_info_line(),
    }
    print(json.dumps(line))


def measure_track_mixed(model, device, batch=32, size=255, lp='fp16', heads_f32=False, seed=0, steps=0, warmup=2,
                        min_seconds=1.0):
    """BASELINE configs[4], one GPU's share: `batch` independent streams in lock step, low-precision
    backbone (+ big head convs), fp32 depthwise xcorr / reduce / predictions, N_q = 7.  Returns
    (plan dict, frames/s fields) — the caller owns barriers for the multi-GPU form."""
    e = model.engine
    t = lambda arr: torch.from_numpy(arr).to(device)
    pr = model.pr_pool
    model.pr_pool = False
    model.template(t(synth.crop(5000 + seed, batch, 127)))
    model.pr_pool = pr
    x = t(synth.crop(6000 + seed, batch, size))
    mem = t(synth.memory_kernels(7000 + seed, 7 * batch))
    sm = torch.ones(batch, 7, device=device)
    dt_ = torch.float16 if lp == 'fp16' else torch.bfloat16
    for _ in range(max(2, warmup)):
        e.track_mixed(x, model.zf, mem, sm, dtype=dt_, heads_lp=not heads_f32)
    return next(v for k, v in e._track.items() if k[:6] == ('mixed', batch, size, 7, dt_, not heads_f32))


FRAME_GFLOP = 54.304194           # one tracked 255^2 frame, N_q = 7 (duplicate search encodes removed)
MIXED_DOMINANT_KERNEL = 'conv_igemm_bf16<256, 256, 4, 4, true, 0, 2, 16, 1>'    # the fp16 256 x 256 tile (scripts/pmc_busy.py key)


def mixed_roofline(pm, batch, size, n, dt, top=5):
    """Roofline object of configs[4]'s per-GPU step (`batch` streams in one mixed-precision plan): algorithmic FLOPs of
    SURVEY 8(d) per tracked frame x batch over the replay time against the dense fp16 MFMA peak; the clock / matrix-pipe
    utilisation of its dominant tile and the step's HBM traffic from the committed counter passes of THIS workload
    (scripts/round_snapshot.sh: `--workload track_mixed`; null when they were measured on another source tree); per-op
    HIP-event spans (plan.profile) for the slowest launches."""
    prof = stable_profile(pm['plan'], dt / n * 1e3, frames=5)
    convs = iter(pm['log'])
    rows, ms_all, ms_conv, fl_conv = [], 0.0, 0.0, 0.0
    for kind, tile, ks, groups, ms in prof:
        ms_all += ms
        if kind in (0, 11, 18, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31):
            try:
                name, M, N, K, g, macs = next(convs)
            except StopIteration:
                continue
            ms_conv += ms
            fl_conv += 2.0 * macs
            rows.append((ms, name, M, N, K, 2.0 * macs / ms / 1e9))
        else:
            rows.append((ms, {1: 'stem', 2: 'maxpool', 3: 'groupdw (9 samples per stream)', 4: 'conf_fusion reduce', 6: 'permute', 10: 'rows copy', 12: 'convert',
                              13: 'maxpool (lp)', 14: 'stem + pool (lp)', 15: 'rows copy', 16: 'prediction convs (thin)'}.get(kind, 'op kind %d' % kind), 0, 0, 0, 0.0))
    gflop = batch * FRAME_GFLOP * (size / 255.0) ** 2 if size != 255 else batch * FRAME_GFLOP
    ach = gflop * n / dt / 1e3
    S = (size - 7) // 2 + 1                                     # stem rows -> response size: 25 at 255, 27 at 271
    S = ((S - 1) // 2 + 1 - 1) // 2 + 1 - 6
    # GroupDW: per sample the three fp32 search maps + kernels in, the response map out in the storage type (3 161 088 B
    # with an fp32 output); Conf_Fusion's reduction: the 7 confidence | value maps in, one map out (storage type)
    gdw = 9 * batch * (GROUPDW_BYTES_PER_SAMPLE - S * S * 256 * 2) if size == 255 else 0
    red = batch * S * S * (7 * 512 + 256) * 2
    alg_bytes = sum(pm.get('lp_bytes', [])) + sum(pm.get('f32_bytes', [])) + batch * 3 * size * size * 4 + gdw + red
    traffic, tsrc, by_kernel = None, None, None
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic_mixed.json')
    if size == 255 and os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        ok, tsrc = counters_current(tj.get('_meta'), 'profiles/pmc_traffic_mixed.json')
        if ok and '--batch %d ' % batch not in tj.get('_meta', {}).get('command', ''):
            ok, tsrc = False, 'profiles/pmc_traffic_mixed.json was measured on another batch size (%s)' % tj.get('_meta', {}).get('command')
        if ok:
            traffic = tj.get('hbm_bytes_per_step')
            tsrc += ' (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, summed over the step)'
            by_kernel = {k: v['hbm_bytes_per_launch'] for k, v in sorted(tj.get('by_kernel', {}).items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches_per_step'])[:6]}
    return {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / BF16_PEAK_TFLOPS, 4),
            **busy_fields(pmc_busy(MIXED_DOMINANT_KERNEL, BF16_PEAK_TFLOPS), ach),
            'algorithmic_gflop_per_step': round(gflop, 1),
            'traffic': traffic, 'traffic_source': tsrc,
            'algorithmic_bytes_per_step': int(alg_bytes),
            'algorithmic_bytes_note': 'every conv launch: operands + result once (storage type of each), the fp32 crops, GroupDW (9 samples per stream) and the Conf_Fusion reduction; layout permutes and prediction maps not counted',
            'traffic_to_algorithmic': round(traffic / alg_bytes, 3) if traffic else None,
            'traffic_per_launch_top_kernels': by_kernel,
            'kernel': 'conv_igemm_bf16<f16> family + heads (%d launches per step)' % len(prof),
            'conv_ms_per_step': round(ms_conv, 3), 'all_ops_ms_per_step': round(ms_all, 3),
            'conv_tflops': round(fl_conv / (ms_conv * 1e-3) / 1e12, 1) if ms_conv else None,
            'slowest': [{'op': nm, 'M': M, 'N': N, 'K': K, 'ms': round(ms, 3), 'tflops': round(tf, 1)}
                        for ms, nm, M, N, K, tf in sorted(rows, reverse=True)[:top]]}


def measure_lockstep_f32(model, device, batch=4, size=255, steps=0, min_seconds=1.0):
    """fp32 lock-step throughput mode: `batch` independent streams batched into ONE fp32 plan (same
    kernels and the same 1e-4 parity as configs[1]; M = batch x 961 pixels fills the chip where one
    stream cannot), N_q = 7."""
    e = model.engine
    t = lambda arr: torch.from_numpy(arr).to(device)
    pr = model.pr_pool
    model.pr_pool = False
    model.template(t(synth.crop(5100, batch, 127)))
    model.pr_pool = pr
    x = t(synth.crop(6100, batch, size))
    mem = t(synth.memory_kernels(7100, 7 * batch))
    sm = torch.ones(batch, 7, device=device)
    for _ in range(2):
        e.track(x, model.zf, template_mem=mem, score_mem=sm)
    p = e._track[(batch, size, 7)]
    n, dt = _timed(p['plan'].run, min_seconds, steps)
    return {'workload': 'fp32 lock-step: %d streams in one fp32 plan (backbone + heads, N_q=7), hipGraph replay, inputs '
                        'resident' % batch,
            'value': round(batch * n / dt, 1), 'unit': 'frames/s', 'batch': batch, 'steps': n,
            'ms_per_step': round(dt / n * 1e3, 4), 'dtype': f32_dtype_label(e)}


def measure_track_271(model, device, warmup=30, min_seconds=1.0):
    """The headline frame at the reference's other instance size (271 x 271 search crops, 27 x 27 response: SURVEY 8f rank 4,
    experiments/test/USOT.yaml's online-update sizes): same per-frame device path and loop as `value`."""
    sess, crops, p = open_stream(model, device, seed=900, size=271)
    conf = Confidences()
    run_frames(sess, crops, p, conf, warmup)
    n, dt = _timed(lambda: run_frames(sess, crops, p, conf, 1), min_seconds)
    return {'workload': 'configs[1] at instance size 271: batch=1 fp32 frame (backbone + heads N_q=7 + decode + PrRoIPool), 27x27 response',
            'value': round(n / dt, 1), 'unit': 'frames/s', 'steps': n, 'ms_per_step': round(dt / n * 1e3, 4), 'dtype': f32_dtype_label(model.engine),
            'search': 271}


def measure_track_split16(model, device, size=255, warmup=30, min_seconds=1.5):
    """The headline frame on the OPT-IN split-fp16 products (engine option split16_f32: the K >= 1152 convolutions and layer3's
    fused 1x1 pairs form every product from hi + lo fp16 halves of the fp32 operands, fp32 accumulation; same 1e-4 parity bar, narrower
    operands than the reference's fp32 - hence a labelled extra, never `value`).  Same loop as `value`, own engine on the same weights."""
    from usot_amd.engine import Engine
    eng = Engine(model, device, **dict({k: v for k, v in model.engine_options.items() if v is not None and k != 'options'},
                                       options=dict(model.engine_options.get('options') or {}, split16_f32=True)))
    keep, model._engine = model._engine, eng
    try:
        sess, crops, p = open_stream(model, device, seed=0, size=size)
    finally:
        model._engine = keep
    conf = Confidences()
    run_frames(sess, crops, p, conf, warmup)
    n, dt = _timed(lambda: run_frames(sess, crops, p, conf, 1), min_seconds)
    return {'workload': 'configs[1] with split-fp16 products (opt-in engine option split16_f32): batch=1 frame, same loop as `value`',
            'value': round(n / dt, 1), 'unit': 'frames/s', 'steps': n, 'ms_per_step': round(dt / n * 1e3, 4), 'dtype': f32_dtype_label(eng),
            'range_fallbacks': getattr(eng, 'range_fallbacks', 0), 'roofline': roofline(sess, frames=10)}


def track_mixed(a, rank, world, device):
    """`--workload track_mixed`: BASELINE configs[4] as its own JSON line (fp16 backbone + fp32 xcorr,
    `--batch` independent streams per GPU: 32 per GPU = 256 on 8 GPUs).  One step = one batched frame
    on every rank."""
    model, wbytes = build_model(rank, world, device)
    b = a.batch
    p = measure_track_mixed(model, device, b, a.size, a.lp, a.heads_f32, seed=rank, warmup=a.warmup)
    steps = agree_steps(p['plan'].run, a.steps, a.min_seconds, device)
    streams.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        p['plan'].run()
    torch.cuda.synchronize()
    streams.barrier()
    dt = streams.max_over_ranks(time.perf_counter() - t0, device=device)
    if rank == 0:
        print(json.dumps({
            'metric': 'tracker FPS (255x255 search, ResNet-50), %s backbone%s + fp32 xcorr, batch %d per GPU'
                      % (a.lp, '' if a.heads_f32 else ' and head convs', b),
            'value': round(world * b * steps / dt, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': steps,
            'steps_requested': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.lp + '+f32', 'data': 'synthetic',
            'config': {'workload': 'configs[4]: %s backbone + fp32 xcorr mixed precision, batch=%d per GPU x %d GPUs, N_q=7'
                                   % (a.lp, b, world), 'search': a.size, 'hipgraph': True,
                       'weights': 'synthetic seed 0 (calibrated BN), %s' % streams.broadcast_summary(),
                       'head_convs': 'f32' if a.heads_f32 else a.lp + ' (encoders, conf/value, towers); preds, memory-kernel encoders, GroupDW, reduce f32'},
            'roofline': mixed_roofline(p, b, a.size, steps, dt), 'build_info': build_info_line()}))
    streams.barrier()


def agree_steps(run, steps, min_seconds, device, probe=5):
    """Number of timed steps: the requested K, raised so that the timed region lasts at least
    `min_seconds` (a 20-step window of a 0.9 ms frame is 18 ms: nothing can be sampled in it).  The
    estimate is the max over ranks, so every rank times the same number of steps."""
    if min_seconds <= 0:
        return int(steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(probe):
        run()
    torch.cuda.synchronize()
    est = streams.max_over_ranks((time.perf_counter() - t0) / probe, device=device)
    return max(int(steps), int(math.ceil(min_seconds / max(est, 1e-6))))


def self_launch(a):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves (one process per GPU,
    RCCL through torch.distributed) and pass rank 0's JSON line through.  The reference fans out with
    mpiexec (scripts/test_epochs_usot.py:19-49)."""
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def plumbing_only(a):
    """`--plumbing-only`: the N-rank control flow of the default workload with the frames left out - what an 8-GPU node would do
    around the timed region, runnable on CPU ranks over gloo: torchrun environment -> streams.init -> the model on rank 0 only ->
    ONE flat broadcast -> stream s on rank s mod N -> barrier / timed no-op / barrier -> max over ranks -> rank 0 prints one line.
    `value` is null: nothing was measured."""
    rank, local, world = streams.env_world()
    if world != a.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    os.environ.setdefault('USOT_ALLOW_GLOO_ON_GPUS', '1')
    streams.init(backend='gloo')
    torch.set_num_threads(min(host_threads(), streams.host_thread_cap(world)))
    model, wbytes = build_model(rank, world, torch.device('cpu'))
    S = max(1, a.streams_per_gpu)
    mine = streams.shard(range(world * S), rank, world)          # the videos this rank would track
    # every rank's model must now hold rank 0's weights: one checksum per rank, gathered through the same max-over-ranks path
    ck = float(sum(float(v.double().sum()) for v in model.state_dict().values() if v.is_floating_point()))
    spread = streams.max_over_ranks(ck) + streams.max_over_ranks(-ck)        # max - min over ranks: 0 when all equal
    streams.barrier()
    t0 = time.perf_counter()
    streams.barrier()
    dt = streams.max_over_ranks(time.perf_counter() - t0)
    counts = [len(streams.shard(range(world * S), r, world)) for r in range(world)]
    if rank == 0:
        print(json.dumps({
            'metric': 'tracker FPS (255x255 search, ResNet-50)', 'value': None, 'unit': 'frames/s', 'n_gpus': world, 'steps': 0,
            'warmup': 0, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic', 'plumbing_only': True,
            'config': {'workload': 'PLUMBING ONLY - no frames run, nothing measured: rendezvous, weight broadcast, sharding, barriers',
                       'streams': world * S, 'streams_per_gpu': S, 'shards': counts, 'rank0_streams': mine,
                       'weights': 'synthetic seed 0 (calibrated BN), %s' % streams.broadcast_summary(), 'weight_bytes': wbytes,
                       'weight_checksum_spread_over_ranks': spread, 'host_threads_per_rank': torch.get_num_threads(),
                       'barrier_seconds': round(dt, 6)}}))
    streams.barrier()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--size', type=int, default=255)
    ap.add_argument('--min-seconds', type=float, default=2.0,
                    help='the timed region lasts at least this long: steps = max(--steps, ceil(min_seconds / step time)); 0 = exactly --steps')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the configs[2]/[4]/lock-step sub-objects of the default line')
    ap.add_argument('--no-xcorr', action='store_true')
    ap.add_argument('--workload', default='track', choices=['track', 'backbone_bf16', 'track_mixed'])
    ap.add_argument('--lp', default='fp16', choices=['fp16', 'bf16'])
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--heads-f32', action='store_true', help='track_mixed: keep every head conv in fp32')
    ap.add_argument('--streams-per-gpu', type=int, default=1,
                    help='independent videos per GPU on separate HIP streams (default 1 = BASELINE configs[1])')
    ap.add_argument('--plumbing-only', action='store_true',
                    help='NO frames are run and nothing is measured: the N-rank control flow only (rendezvous, one flat weight broadcast, stream '
                         'sharding, barriers, max-over-ranks, rank 0 prints one line) - runs without a GPU over gloo (tests/test_distributed_cpu.py)')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(a)
    if a.plumbing_only:
        return plumbing_only(a)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the HIP path)')
    ndev = torch.cuda.device_count()
    rank, local, world = streams.env_world()
    if world != a.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    # one rank per GPU; when fewer GPUs are visible than ranks (a 1-GPU box running the N-rank path as a
    # functional test) ranks share devices and the weight broadcast falls back to gloo — RCCL refuses two
    # ranks on one device
    oversub = world > ndev
    device = torch.device('cuda', local % ndev)
    torch.cuda.set_device(device)
    streams.init(backend='gloo' if oversub else None, device_index=device.index)
    # the box shows 256 hardware threads under a 16-CPU quota; N ranks share that quota: cap every rank at quota // N
    torch.set_num_threads(min(host_threads(), streams.host_thread_cap(world)))
    if a.workload == 'backbone_bf16':
        if rank == 0:
            backbone_bf16(a, device)
        return
    if a.workload == 'track_mixed':
        track_mixed(a, rank, world, device)
        return

    model, wbytes = build_model(rank, world, device)
    S = max(1, a.streams_per_gpu)
    group = []
    for k in range(S):
        sess, crops, p = open_stream(model, device, seed=rank * S + k, size=a.size)
        group.append((sess, crops, p, Confidences(), torch.cuda.Stream() if S > 1 else torch.cuda.current_stream()))
    sess, crops, p, conf, _ = group[0]
    go = (lambda n: run_frames(sess, crops, p, conf, n)) if S == 1 else (lambda n: run_frames_multi(group, n))
    go(a.warmup)
    steps = agree_steps(lambda: go(1), a.steps, a.min_seconds, device)

    streams.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(steps)
    torch.cuda.synchronize()
    streams.barrier()
    dt = streams.max_over_ranks(time.perf_counter() - t0, device=device)

    if rank == 0:
        fps = world * S * steps / dt
        line = {
            'metric': 'tracker FPS (255x255 search, ResNet-50)', 'value': round(fps, 2), 'unit': 'frames/s',
            'n_gpus': world, 'steps': steps, 'steps_requested': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            # the engine's default: exact fp32 products (v_mfma_f32_16x16x4_f32), fp32 storage and accumulation - the reference's
            # arithmetic; the opt-in split-fp16 mode is timed as the labelled extra `track_split16` below
            'dtype': f32_dtype_label(model.engine),
            'data': 'synthetic',
            'config': {'workload': 'configs[1]: batch=1 ResNet-50(layer3)+neck, fused depthwise xcorr, cls/reg/'
                                   'memory heads (N_q=7), decode + PrRoIPool, fp32, 1 stream per GPU',
                       'search': a.size, 'template': 127, 'streams': world * S, 'streams_per_gpu': S, 'weights': 'synthetic seed 0 '
                       '(calibrated BN), %s' % streams.broadcast_summary(), 'hipgraph': True,
                       'devices_visible': ndev, 'host_threads_per_rank': torch.get_num_threads(),
                       'sessions': len({id(g[0]) for g in group}),
                       # distinct pinned result blocks and distinct HIP streams: every video has its own
                       'result_streams': len({g[0].out8.data_ptr() for g in group}), 'hip_streams': len({g[4].cuda_stream for g in group})},
        }
        line['roofline'] = roofline(sess, frames=10)

        def extra(key, fn):
            # a sub-measurement that fails must not take the headline line with it: the line then carries the error text under its key
            try:
                line[key] = fn()
            except Exception as exc:      # noqa: BLE001
                line[key] = {'error': '%s: %s' % (type(exc).__name__, str(exc)[:300])}

        def mixed_b32():
            pm = measure_track_mixed(model, device, 32, a.size)
            n, t = _timed(pm['plan'].run, 1.0)
            return {'workload': 'configs[4], one GPU: fp16 backbone + head convs, fp32 xcorr / reduce / predictions, 32 streams in '
                                'lock step, N_q=7, one hipGraph', 'value': round(32 * n / t, 1), 'unit': 'frames/s',
                    'steps': n, 'ms_per_step': round(t / n * 1e3, 3), 'dtype': 'fp16+f32',
                    'roofline': mixed_roofline(pm, 32, a.size, n, t)}
        if not a.no_xcorr:
            extra('xcorr_hbm', lambda: xcorr_bandwidth(device))
        if world == 1:
            extra('video_loop_pcie_inclusive', lambda: video_loop(model, device))
        if world == 1 and not a.no_extras:
            # the two low-precision north-star configurations, the opt-in split-fp16 frame and the fp32 lock-step mode, measured by
            # the same process (each a few seconds; their own full lines: --workload backbone_bf16 / track_mixed)
            extra('backbone_bf16_b64', lambda: measure_backbone_bf16(model, device, 64, a.size))
            extra('track_mixed_b32', mixed_b32)
            extra('track_split16', lambda: measure_track_split16(model, device, a.size))
            extra('lockstep_f32_b4', lambda: measure_lockstep_f32(model, device, 4, a.size))
            if a.size != 271:
                extra('track_271', lambda: measure_track_271(model, device))
        if world == 1 and not a.no_cpu_baseline:
            extra('cpu_baseline', cpu_baseline)
        line['build_info'] = build_info_line()
        print(json.dumps(line))
    streams.barrier()


if __name__ == '__main__':
    main()
