import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, torch.nn.functional as F
from usot_amd import hip
DEV = 'cuda:0'
torch.manual_seed(0)
N, Cin, H, W, Cout = 1, 64, 8, 8, 64
def run(x, w, tile):
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
    y = hip.conv2d(xd, wd, torch.zeros(Cout, device=DEV), KH=1, KW=1, tile=tile)
    return y.permute(0, 3, 1, 2).cpu()
for name, x, w in (('ones', torch.ones(N, Cin, H, W), torch.ones(Cout, Cin, 1, 1)),
                   ('x=k index', torch.arange(Cin).float().view(1, Cin, 1, 1).expand(N, Cin, H, W).contiguous(), torch.ones(Cout, Cin, 1, 1)),
                   ('w=k index', torch.ones(N, Cin, H, W), torch.arange(Cin).float().view(1, Cin, 1, 1).expand(Cout, Cin, 1, 1).contiguous()),
                   ('x=pixel', torch.arange(H * W).float().view(1, 1, H, W).expand(N, Cin, H, W).contiguous(), torch.ones(Cout, Cin, 1, 1)),
                   ('w=co', torch.ones(N, Cin, H, W), torch.arange(Cout).float().view(Cout, 1, 1, 1).expand(Cout, Cin, 1, 1).contiguous()),
                   ('randn', torch.randn(N, Cin, H, W), torch.randn(Cout, Cin, 1, 1))):
    ref = F.conv2d(x, w)
    y = run(x, w, 91)
    print(name, 'ref[0,:4,0,:4]', ref[0, :4, 0, :4].flatten().tolist()[:8], '\n   got', y[0, :4, 0, :4].flatten().tolist()[:8], ' max|diff|', float((y - ref).abs().max()))
print('--- one-hot channel c in x (w = ones): expected 1 everywhere')
res = []
for c in range(Cin):
    x = torch.zeros(N, Cin, H, W); x[:, c] = 1.0
    y = run(x, torch.ones(Cout, Cin, 1, 1), 91)
    res.append(float(y[0, 0, 0, 0]))
print(res)
print('--- x = 3.0 everywhere, then 0.001, then 1000.5')
for val in (3.0, 0.001, 1000.5, 1.0 + 2.0 ** -12):
    y = run(torch.full((N, Cin, H, W), val), torch.ones(Cout, Cin, 1, 1), 91)
    print(val, float(y[0, 0, 0, 0]), 'expected', 64 * val)
print('--- w = 1 + 2^-12 (needs lo), x = 1')
y = run(torch.ones(N, Cin, H, W), torch.full((Cout, Cin, 1, 1), 1.0 + 2.0 ** -12), 91)
print(float(y[0, 0, 0, 0]), 'expected', 64 * (1.0 + 2.0 ** -12))
