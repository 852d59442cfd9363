#!/usr/bin/env python3
"""Where a batch-1 k-step goes: builds the library with -DUSOT_TRACE (s_memtime stamps in one workgroup's
consumer wave 0 and producer wave 0 of conv_igemm_f32_v3 / v4, written to the split-K workspace), runs layer3's
3x3 dilated conv (M = 961, N = 256, K = 2304) on the given tiles and prints the median phase lengths in
shader-clock cycles.   python scripts/trace_kstep.py 41 53 54"""
import ctypes as C, glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# USOT_TRACE_LIB: a library already built with -DUSOT_TRACE (conv_igemm.hip is the only file that reads the macro)
trace_lib = os.environ.get('USOT_TRACE_LIB') or os.path.join(ROOT, 'gpurun_out', 'libusot_hip_trace.so')
if not os.environ.get('USOT_TRACE_LIB'):
    os.makedirs(os.path.dirname(trace_lib), exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(ROOT, 'usot_amd', 'csrc', '*.hip')))
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++20', '-fPIC', '-shared', '-DUSOT_TRACE',
                           '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'usot_amd', 'csrc'), '-o', trace_lib] + srcs)
os.environ['USOT_HIP_LIB'] = trace_lib
import torch
from usot_amd import hip
L = hip.lib(); dev = 'cuda:0'
N, H, W, Cin, Cout, k, pad, dil = 1, 31, 31, 256, 256, 3, 2, 2
x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k * k * Cin, device=dev) * 0.02
b = torch.randn(Cout, device=dev); y = torch.empty(N, H, H, Cout, device=dev)
for tile in [int(v) for v in sys.argv[1:]] or [41, 53]:
    ws = torch.zeros(8 * 64, dtype=torch.int32, device=dev)
    d = hip.conv_desc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N=N, H=H, W=W, Cin=Cin, OH=H, OW=H, Cout=Cout,
                      KH=k, KW=k, pad=(pad, pad), dil=(dil, dil), act=1, tile=tile, ws=ws.data_ptr(), w_frag=hip.tile_wfrag(tile),
                      w_scale=b.data_ptr() if hip.tile_wfrag(tile) == 2 else None, x_split=hip.tile_xsplit(tile))      # timing only: random bits are as good as split filters
    for _ in range(2):
        hip.check(L.usot_conv2d_f32(hip.stream(), C.byref(d)))
    torch.cuda.synchronize()
    t = (ws.cpu().numpy().astype(np.int64) & 0xffffffff).reshape(64, 8)[2:34]
    med = lambda a: int(np.median(a))
    print('%-44s k-step %5d | consumer: MFMA phase %4d, barrier wait %4d | producer: wait for loads %4d, convert + LDS stores %4d, loads %4d, barrier wait %4d' % (
        hip.tile_name(tile), med(np.diff(t[:, 0])), med(t[:, 1] - t[:, 0]), med(t[:, 2] - t[:, 1]),
        med(t[:, 3] - t[:, 4]), med(t[:, 5] - t[:, 3]), med(t[:, 6] - t[:, 5]), med(t[:, 7] - t[:, 6])))
