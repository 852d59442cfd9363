"""CPU, build container only (skipped where /root/reference is absent): the reference's own
scripts/test_usot.py is IMPORTED unchanged with this repo's `lib` on the path — every name it
imports resolves here — and its `track()` driver runs end to end against a stub tracker (the
model itself needs the GPU).  cv2 / easydict are not installed in this image: minimal stand-ins
for the four cv2 calls and the attribute dict the script uses."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

REF_SCRIPT = '/root/reference/scripts/test_usot.py'
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SCRIPT), reason='reference tree not present')


def _load_script(monkeypatch, frames):
    cv2 = types.ModuleType('cv2')
    cv2.imread = lambda path: frames[int(os.path.basename(path).split('.')[0])]
    ticks = iter(range(1, 10 ** 6))
    cv2.getTickCount = lambda: next(ticks)
    cv2.getTickFrequency = lambda: 1.0
    cv2.cvtColor = lambda im, code: im
    cv2.COLOR_GRAY2BGR = 0
    ed = types.ModuleType('easydict')

    class EasyDict(dict):
        __getattr__ = dict.get
        __setattr__ = dict.__setitem__
    ed.EasyDict = EasyDict
    monkeypatch.setitem(sys.modules, 'cv2', cv2)
    monkeypatch.setitem(sys.modules, 'easydict', ed)
    monkeypatch.setattr(sys, 'dont_write_bytecode', True)
    spec = importlib.util.spec_from_file_location('ref_test_usot', REF_SCRIPT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_test_script_runs_on_this_lib(monkeypatch, tmp_path):
    from usot_amd import synth
    frames = [synth.frame(3, t=t)[0] for t in range(4)]
    mod = _load_script(monkeypatch, frames)
    import lib.models.models as models
    assert mod.models is models and mod.USOTTracker.__module__ == 'usot_amd.tracker'
    assert mod.load_pretrain.__module__ == 'usot_amd.io_utils' and mod.load_dataset.__module__ == 'usot_amd.benchmarks'
    net = mod.models.__dict__['USOT']()            # what main() does (test_usot.py:138)
    assert hasattr(net, 'template') and hasattr(net, 'track') and hasattr(net, 'extract_memory_feature')

    class StubTracker:                              # the script only reads these three keys
        def init(self, im, pos, sz, net):
            return {'target_pos': pos, 'target_sz': sz, 'cls_score': 1.0}

        def track(self, state, im):
            state['target_pos'] = state['target_pos'] + 1.0
            return state
    args = types.SimpleNamespace(epoch_test=False, dataset='OTB2015', arch='USOT', resume='x.pth')
    video = {'name': 'synth', 'image_files': ['%d.jpg' % i for i in range(4)],
             'gt': np.array([[100.0, 80.0, 40.0, 30.0]] * 4)}
    monkeypatch.chdir(tmp_path)
    mod.track(StubTracker(), net, video, args)
    out = (tmp_path / 'var' / 'result' / 'OTB2015' / 'USOT' / 'synth.txt').read_text().strip().splitlines()
    assert len(out) == 4 and out[1].split(',')[2:] == ['40.0', '30.0']
