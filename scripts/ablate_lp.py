#!/usr/bin/env python3
"""Where a batch-64 bf16 1x1 expansion conv (+ residual + ReLU; M = 61 504, 256 -> 1024: 283 MB of HBM traffic) spends
its time: the shipped kernel against builds with parts removed (-DUSOT_LPABL_NOMMA / NOLOAD / NORES / NOSTORE), rotating
three sets of buffers so that nothing stays in the Infinity Cache.  Only the durations matter.
    python scripts/ablate_lp.py <lib.so> [tile ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['USOT_HIP_LIB'] = os.path.abspath(sys.argv[1])
import torch
from usot_amd import hip
dev = 'cuda:0'
M, Cin, Cout = 64 * 31 * 31, 256, 1024
sets = []
for _ in range(3):
    sets.append((torch.randn(64, 31, 31, Cin, device=dev).to(torch.bfloat16), torch.randn(64, 31, 31, Cout, device=dev).to(torch.bfloat16)))
w = (torch.randn(Cout, Cin, device=dev) * 0.05).to(torch.bfloat16); b = torch.randn(Cout, device=dev)
out = []
for tile in [int(v) for v in sys.argv[2:]] or [25, 21, 24]:
    run = lambda i: hip.conv2d_bf16(sets[i % 3][0], w, b, KH=1, KW=1, res=sets[i % 3][1], act=hip.ACT_RELU, tile=tile)
    for i in range(6): run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(60): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 60 * 1e3
    out.append('tile %d: %.1f us (%.2f TB/s of the 283 MB)' % (tile, us, M * (Cin * 2 + Cout * 4) / us / 1e6))
print('%-26s %s' % (os.path.basename(sys.argv[1]), ' | '.join(out)))
