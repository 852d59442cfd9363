#!/usr/bin/env python3
"""Builds usot_amd/csrc/alt/lib_<name>.so: the library with ONE translation unit recompiled under extra -D flags (timing /
ablation builds; usot_amd/csrc/alt/ is git-ignored but travels to the GPU box).  The other objects are the up-to-date ones
of the normal build.    python scripts/build_variant.py <name> <file.hip> -DUSOT_ABL_NOW [...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usot_amd import build as b
name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build()
alt = os.path.join(b.CSRC, 'alt')
os.makedirs(alt, exist_ok=True)
obj = os.path.join(alt, '%s_%s.o' % (name, unit[:-4]))
subprocess.check_call([b._hipcc()] + b.FLAGS + b.FILE_FLAGS.get(unit, []) + flags + ['-c', os.path.join(b.CSRC, unit), '-o', obj])
objs = [obj if os.path.basename(s) == unit else s[:-4] + '.o' for s in b.sources()]
lib = os.path.join(alt, 'lib_%s.so' % name)
subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
print(lib)
