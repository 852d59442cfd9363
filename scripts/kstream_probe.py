#!/usr/bin/env python3
"""How fast is the accumulator-stationary 1x1 kernel (csrc/pw_kstream.hip) when HBM is out of the way?  M = 8 192 (32 panels on
32 CUs, X = 16 MB: cache-resident on repeat) gives the time of ONE panel = 16 chunks of 64 MFMAs per wave; M = 65 536 one
panel on every CU with X (134 MB) in the Infinity Cache; M = 61 504 x 3 rotating buffers the HBM-bound case of the backbone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usot_amd import hip
dev = 'cuda:0'
K, N = 1024, 256
w = (torch.randn(N, K, device=dev) / 32).bfloat16(); b = torch.randn(N, device=dev)
for M, nbuf in ((8192, 1), (65536, 1), (61504, 3)):
    xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nbuf)]
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run = lambda i: hip.check(hip.lib().usot_pw_kstream_lp(hip.stream(), hip.ptr(xs[i % nbuf]), hip.ptr(w), hip.ptr(b), hip.ptr(y), M, K, N, 1, 0), 'ks')
    for i in range(6): run(i)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for i in range(50): run(i)
    e[1].record(); torch.cuda.synchronize()
    us = e[0].elapsed_time(e[1]) / 50 * 1e3
    panels = (M + 255) // 256
    print('M = %6d (%3d panels, %d buffers): %6.1f us per launch, %6.1f TFLOP/s, %5.2f TFLOP/s per busy CU (7.3 = the matrix pipe at 1.9 GHz)'
          % (M, panels, nbuf, us, 2.0 * M * K * N / us / 1e6, 2.0 * M * K * N / us / 1e6 / min(panels, 256)), flush=True)
