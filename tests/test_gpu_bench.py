"""bench.py's N > 1 branch on the GPU box the driver tests on: until an 8-GPU node is available the
multi-rank path — self-launch under torch.distributed.run, one rank per stream, rank 0's weights reaching every rank through
ONE flat broadcast, barrier + max-over-ranks timing, rank 0 printing the one JSON line — runs as TWO ranks sharing the one
visible GPU (the gloo-oversubscribed branch of bench.py: RCCL refuses two ranks on one device).  The reference's fan-out is
scripts/test_epochs_usot.py:19-49 (mpiexec, no message at all)."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHT_BYTES = (29414993 + 44486 - 70) * 4          # every parameter + float BN buffer, fp32 (tests/test_distributed_cpu.py)


def _bench(*args, timeout=600):
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)                                   # bench.py must start its own ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert r.returncode == 0, r.stderr.decode(errors='replace')[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line (rank 0), got %d' % len(lines)
    return json.loads(lines[0])


def test_bench_two_ranks_shard_streams_and_broadcast_weights():
    j = _bench('--gpus', '2', '--steps', '50', '--warmup', '5', '--min-seconds', '0', '--no-extras', '--no-xcorr', '--no-cpu-baseline')
    assert j['n_gpus'] == 2 and j['steps'] == 50 and j['warmup'] == 5 and j['scaling'] == 'weak'
    assert j['config']['streams'] == 2 and j['config']['streams_per_gpu'] == 1
    assert 'broadcast %d B in ' % WEIGHT_BYTES in j['config']['weights'] and 'backend gloo, 2 ranks' in j['config']['weights'], j['config']['weights']
    assert j['config']['host_threads_per_rank'] >= 1
    assert j['unit'] == 'frames/s' and math.isfinite(j['value']) and j['value'] > 0
    # value = frames of ALL ranks / max-over-ranks time: consistent with the per-step time the line reports
    assert abs(j['value'] - 2 * 1e3 / j['ms_per_step']) <= 0.01 * j['value']
    assert j['roofline']['bound'] == 'mfma' and 0 < j['roofline']['frac'] < 1
    assert 'cpu_baseline' not in j and 'video_loop_pcie_inclusive' not in j       # rank 0 at N = 1 only


def test_bench_line_carries_the_contract_fields():
    """One rank, the short form of the driver's command: every key the contract names, the roofline object with its
    algorithmic bytes beside the measured traffic, and the 271 x 271 sub-object."""
    j = _bench('--steps', '40', '--warmup', '5', '--min-seconds', '0', '--no-xcorr', '--no-cpu-baseline')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline'):
        assert k in j, k
    r = j['roofline']
    # the headline is the EXACT-fp32 frame (BASELINE configs[1] is fp32: v_mfma_f32_16x16x4_f32, peak 157.3 TFLOP/s); the opt-in
    # split-fp16 frame (three fp16 MFMAs per product block: peak = 2500 / 3) is a labelled extra of the same line and says what it is
    assert j['n_gpus'] == 1 and j['dtype'] == 'f32' and j['vs_baseline'] is None
    assert r['peak'] == 157.3 and r['arithmetic'].startswith('v_mfma_f32_16x16x4_f32'), r
    assert abs(r['executed_tflops'] - r['achieved']) < 0.1
    sp = j['track_split16']
    assert 'error' not in sp and sp['dtype'].startswith('f32 storage / accumulation, split-fp16 products') and sp['range_fallbacks'] == 0
    assert sp['roofline']['peak'] == 833.3 and sp['roofline']['arithmetic'].startswith('split fp16') and sp['value'] > j['value']
    assert j['lockstep_f32_b4']['dtype'] == 'f32' and j['track_271']['dtype'] == 'f32'
    for key in ('backbone_bf16_b64', 'track_mixed_b32', 'lockstep_f32_b4', 'track_271', 'video_loop_pcie_inclusive'):
        assert 'error' not in j[key], (key, j[key])
    lp0 = j['backbone_bf16_b64']
    # 64 x 28.192642 GFLOP over the STEP time (not over the sum of the conv launches' spans)
    assert abs(lp0['roofline']['achieved'] - 64 * 28.192642 / lp0['ms_per_step']) < 2.0, lp0['roofline']
    assert r['algorithmic_bytes_per_launch'] > 0
    if r['traffic'] is not None:
        assert r['traffic_to_algorithmic'] >= 1.0, r                              # HBM traffic below the compulsory bytes is a bookkeeping error
    lp = j['backbone_bf16_b64']['roofline']
    if lp['traffic'] is not None:
        assert lp['traffic_to_algorithmic'] >= 1.0, lp
    assert j['track_271']['search'] == 271 and j['track_271']['value'] > 0


def test_eight_videos_on_one_gpu_are_eight_sessions_and_beat_one_stream():
    """configs[3]'s eight independent videos on the ONE GPU the driver has (`--streams-per-gpu 8`): eight sessions with their
    own result blocks and HIP streams, frames of different videos overlapping on the device - the aggregate rate must clearly
    beat the single-stream rate of the same process kind (measured 1.3-1.4 x at 4-8 videos; the bar is 1.2 x)."""
    common = ['--steps', '150', '--warmup', '20', '--min-seconds', '1', '--no-extras', '--no-xcorr', '--no-cpu-baseline']
    one = _bench(*common)
    eight = _bench('--streams-per-gpu', '8', *common)
    c = eight['config']
    assert c['streams'] == 8 and c['streams_per_gpu'] == 8 and c['sessions'] == 8
    assert c['result_streams'] == 8 and c['hip_streams'] == 8, c
    assert one['config']['sessions'] == 1
    assert eight['value'] >= 1.2 * one['value'], (eight['value'], one['value'])
    # value counts the frames of ALL videos: steps x 8 / time
    assert abs(eight['value'] - 8 * 1e3 / eight['ms_per_step']) <= 0.01 * eight['value']
