#!/usr/bin/env python3
"""Prints the float64 acceptance table (tests/golden/f64_gate.py) for the current build on cuda:0, both weight families."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import f64_gate
from usot_amd import synth
from usot_amd.model import USOT
gold = f64_gate.load()
for fam in synth.FAMILIES:
    m = USOT(); m.load_state_dict(synth.torch_state_dict(m, seed=0, calibrated=True, family=fam), strict=True); m.eval(); m = m.to('cuda:0')
    rows = f64_gate.table(gold, fam, f64_gate.run_model(m))
    print(f64_gate.fmt(fam, rows))
    print('violations:', f64_gate.violations(rows))
