import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'experiments: parity of EXPERIMENTAL conv tiles / kernels no tuning table routes; they exist only in a '
                                       'library built with USOT_EXPERIMENTS=1 (usot_amd/build.py) and are skipped on the default one')


def _lib():
    import ctypes
    from usot_amd import build
    return ctypes.CDLL(build.LIB) if os.path.exists(build.LIB) else None


def tile_is_built(tile, lp=False):
    """Whether conv tile id `tile` (fp32 table, or the low-precision table with lp=True) is compiled into the in-tree library: the
    routed tiles always, the experimental ids only in a USOT_EXPERIMENTS=1 build.  Host-side query: usable at collection time."""
    L = _lib()
    if L is None or tile == 0:
        return True
    return bool((L.usot_conv_bf16_tile_built if lp else L.usot_conv_tile_built)(int(tile)))


def experiments_built():
    L = _lib()
    return bool(L is not None and L.usot_experiments_built())


def tile_params(tiles, lp=False):
    """pytest params for a list of tile ids: ids the default library does not hold carry the `experiments` marker."""
    return [pytest.param(t, marks=[] if tile_is_built(t, lp) else [pytest.mark.experiments]) for t in tiles]


def pytest_collection_modifyitems(config, items):
    if experiments_built():
        return
    skip = pytest.mark.skip(reason='experimental tile / kernel: only in a USOT_EXPERIMENTS=1 build of libusot_hip.so')
    for it in items:
        if 'experiments' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def gold_model():
    with np.load(os.path.join(GOLD, 'golden_model.npz')) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def gold_host():
    with np.load(os.path.join(GOLD, 'golden_host.npz')) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def oracle_sd():
    """Calibrated synthetic state dict (torch CPU tensors) keyed like the reference's."""
    import json
    import torch
    from usot_amd import synth
    with open(os.path.join(GOLD, 'state_dict_keys.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = synth.make_state_dict(shapes, seed=0, calibrated=True)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
