import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scripts')
import torch
from usot_amd import hip
dev='cuda:0'
def run(name, N, H, Cin, Cout, k, pad, tiles, groups=1, dil=(1,1), reps=20, f16=True):
    dt = torch.float16 if f16 else torch.bfloat16
    x = torch.randn(groups, N, H, H, Cin, device=dev).to(dt)
    w = (torch.randn(groups*Cout, k*k*Cin, device=dev)*0.02).to(dt); b = torch.randn(groups*Cout, device=dev)
    out=[]
    for tile in tiles:
        f = lambda: hip.conv2d_bf16(x if groups>1 else x[0], w, b, KH=k, KW=k, pad=(pad,pad), dil=dil, act=hip.ACT_RELU, tile=tile, groups=groups, out_f32=True)
        try:
            for _ in range(3): f()
        except Exception as e:
            out.append('%d:err' % tile); continue
        torch.cuda.synchronize(); best=1e9
        for rep in range(3):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)/reps*1e3)
        out.append('%d:%.1f' % (tile,best))
    print(name,' '.join(out),flush=True)
T=[1,4,5,10,11,12,13,14,25,37,23,19,32,18]
run('enc_k0 M=5600', 224, 7, 256, 256, 3, 0, T)
run('enc_k1 M=3360 d(2,1)', 224, 7, 256, 256, 3, 0, [4,5,11,12,13,14,25], dil=(2,1))
run('tower M=20000 g=3', 32, 25, 256, 256, 3, 1, T, groups=3)
run('enc_s M=26912', 32, 31, 256, 512, 3, 0, [32,21,18,19,23,25,37])
