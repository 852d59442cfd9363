"""y_split -> x_split round trip: conv A (PF = 5 tile, y_split) feeds conv B (PF = 6 tile) without an fp32 map in between."""
import sys
sys.path.insert(0, '/root/repo')
import ctypes as C
import numpy as np, torch, torch.nn.functional as F
from usot_amd import hip
DEV = 'cuda:0'
g = torch.Generator().manual_seed(1)
N, Cc, H, W = 3, 256, 25, 25
x = torch.randn(N, Cc, H, W, generator=g).abs()
w1 = torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc); b1 = torch.randn(Cc, generator=g)
w2 = torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(9 * Cc); b2 = torch.randn(Cc, generator=g)
r1 = F.relu(F.conv2d(x.double(), w1.double(), b1.double(), 1, 1))
r2 = F.relu(F.conv2d(r1, w2.double(), b2.double(), 1, 1)).float()
pk = lambda w: w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous().to(DEV)
xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
y1 = hip.conv2d(xd, pk(w1), b1.to(DEV), KH=3, KW=3, pad=(1, 1), act=hip.ACT_RELU, tile=106, y_split=True)
e1 = (hip.unsplit_map(y1).permute(0, 3, 1, 2).cpu() - r1.float()).abs().max() / r1.abs().max()
# feed the split map straight into the all-DMA tile (bypass conv2d's own conversion)
for tile in (116, 117):
    w2p, sc = hip.split16_pack(pk(w2))
    y2 = torch.empty(N, H, W, Cc, device=DEV)
    d = hip.conv_desc(y1.data_ptr(), w2p.data_ptr(), b2.to(DEV).data_ptr(), y2.data_ptr(), N=N, H=H, W=W, Cin=Cc, OH=H, OW=W, Cout=Cc, KH=3, KW=3,
                      pad=(1, 1), act=hip.ACT_RELU, tile=tile, w_frag=2, w_scale=sc.data_ptr(), x_split=1)
    bb = b2.to(DEV); d.bias = bb.data_ptr()
    hip.check(hip.lib().usot_conv2d_f32(hip.stream(), C.byref(d)), 'conv')
    torch.cuda.synchronize()
    e2 = (y2.permute(0, 3, 1, 2).cpu() - r2).abs().max() / r2.abs().max()
    print('tile %d: level 1 (y_split) err %.2e, level 2 (x_split) err %.2e' % (tile, float(e1), float(e2)))
