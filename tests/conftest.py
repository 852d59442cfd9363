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
