import cProfile, pstats, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from usot_amd import synth
from usot_amd.model import USOT
from usot_amd.tracker import USOTTracker
class Info: arch = 'USOT'
m = USOT(); m.load_state_dict(synth.torch_state_dict(m)); m.eval(); m = m.cuda()
trk = USOTTracker(Info())
ims = [np.ascontiguousarray(synth.frame(77, t=t)[0]) for t in range(16)]
im0, (cx, cy) = synth.frame(77, t=0)
state = trk.init(im0, np.array([cx, cy]), np.array([52.0, 38.0]), m)
for i in range(10): state = trk.track(state, ims[i % 16])
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for i in range(100): state = trk.track(state, ims[i % 16])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
pr.disable()
print('fps', 100 / dt)
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
