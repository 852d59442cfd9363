"""CPU, world size 2 over gloo: the N>1 path of bench.py — rank 0's weights reach every
rank through ONE flat broadcast, streams shard by rank, and the only other collective is
the max over ranks of the timing scalar."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      USOT_ALLOW_GLOO_ON_GPUS='1')      # this test asks for gloo explicitly, also on a box with >= 2 GPUs
    from usot_amd import streams, synth
    from usot_amd.model import USOT
    r, _, w = streams.init(backend='gloo')
    assert (r, w) == (rank, world)
    m = USOT()
    if rank == 0:
        m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True), strict=True)
    nbytes = streams.broadcast_weights(m, src=0)
    sd = m.state_dict()
    want = synth.torch_state_dict(m, seed=0, calibrated=True)
    same = all(torch.equal(sd[k].reshape(-1), want[k].reshape(-1).to(sd[k].dtype)) for k in want)
    t = streams.max_over_ranks(1.0 + rank)
    q.put((rank, nbytes, same, streams.shard(range(8), rank, world), t))
    streams.barrier()
    dist.destroy_process_group()


def test_broadcast_and_shard_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nbytes, same, mine, t in out:
        assert same, 'rank %d did not receive rank 0 weights' % rank
        assert nbytes == (29414993 + 44486 - 70) * 4       # params + BN float buffers (70 int counters apart)
        assert mine == list(range(8))[rank::2]
        assert t == 2.0


def test_single_process_is_a_noop():
    from usot_amd import streams
    from usot_amd.model import USOT
    assert streams.broadcast_weights(USOT(), src=0) == 0
    assert streams.shard('abcdef', 1, 3) == ['b', 'e']
    assert streams.max_over_ranks(3.5) == 3.5


def test_host_thread_cap_divides_the_quota_between_ranks():
    """host cores are the one resource streams on different GPUs share; a rank gets quota // world threads."""
    from usot_amd import streams
    assert streams.host_thread_cap(8, quota=16) == 2
    assert streams.host_thread_cap(1, quota=16) == 16
    assert streams.host_thread_cap(8, quota=4) == 1
    assert streams.host_thread_cap(2) >= 1


def test_init_refuses_gloo_when_every_rank_has_its_own_gpu(monkeypatch):
    """The RCCL branch is the product path on a GPU box: a gloo group with world <= visible GPUs raises instead of silently
    moving the weights through host memory (the 2-ranks-on-1-GPU functional run, world > devices, stays allowed)."""
    import torch
    from usot_amd import streams
    monkeypatch.setenv('RANK', '0'); monkeypatch.setenv('LOCAL_RANK', '0'); monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(torch.distributed, 'is_initialized', lambda: False)
    called = []
    monkeypatch.setattr(torch.distributed, 'init_process_group', lambda **kw: called.append(kw))
    import pytest
    with pytest.raises(RuntimeError, match='must broadcast over RCCL'):
        streams.init(backend='gloo')
    assert not called
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)       # oversubscribed functional run: allowed
    monkeypatch.setattr(torch.distributed, 'get_backend', lambda: 'gloo')
    streams.init(backend='gloo')
    assert called and called[0]['backend'] == 'gloo'


def test_bench_gpus_8_control_flow_over_gloo():
    """`python bench.py --gpus 8` on CPU ranks (the reference's fan-out is scripts/test_epochs_usot.py:19-49): the
    self-launch under torch.distributed.run, eight ranks over gloo, ONE flat broadcast of rank 0's weights (every parameter + float
    BN buffer, fp32), stream s on rank s mod 8, barriers, max-over-ranks, exactly one JSON line from rank 0.  `--plumbing-only`:
    no frame is run and the line says so (`value` null) - the GPU work of the same command is covered by tests/test_gpu_bench.py."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, USOT_ALLOW_GLOO_ON_GPUS='1', OMP_NUM_THREADS='1', CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--plumbing-only'], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors='replace')[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    c = j['config']
    assert j['n_gpus'] == 8 and j['plumbing_only'] is True and j['value'] is None and j['scaling'] == 'weak'
    assert c['streams'] == 8 and c['shards'] == [1] * 8 and c['rank0_streams'] == [0]
    assert c['weight_bytes'] == (29414993 + 44486 - 70) * 4 == 117837636
    assert 'broadcast 117837636 B in ' in c['weights'] and 'backend gloo, 8 ranks' in c['weights'], c['weights']
    assert c['weight_checksum_spread_over_ranks'] == 0.0          # every rank ended up with rank 0's weights
