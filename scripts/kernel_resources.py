#!/usr/bin/env python3
"""VGPRs / spills / occupancy of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kernel_resources.py conv_igemm.hip [name filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import build as b
unit, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
cmd = [b._hipcc()] + b.FLAGS + b.FILE_FLAGS.get(unit, []) + ['-c', os.path.join(b.CSRC, unit), '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    if flt in name:
        short = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0]
        print('%-70s VGPR %3d AGPR %3d spill %3d SGPR %3d occ %d' % (short[:70], r.get('VGPRs', -1), r.get('AGPRs', -1), r.get('VGPRs Spill', -1),
                                                                 r.get('TotalSGPRs', -1), r.get('Occupancy', -1)))
