"""Builds usot_amd/csrc/libusot_hip.so for gfx950 with hipcc (in-tree, no JIT cache)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB = os.path.join(CSRC, 'libusot_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + INCLUDE, '-I' + CSRC]
# USOT_EXPERIMENTS=1 in the environment: compile the experimental conv tiles too (csrc/conv_igemm.hip / conv_bf16.hip: the ids no
# tuning table routes - lab notebook A.3); the default library holds the routed tiles only.  Part of csrc_tree(): the two builds
# never share counters or objects silently.
EXPERIMENTS = os.environ.get('USOT_EXPERIMENTS', '0') == '1'
if EXPERIMENTS:
    FLAGS = FLAGS[:4] + ['-DUSOT_EXPERIMENTS'] + FLAGS[4:]
# per-file flags.  xcorr.hip: hipcc's SLP vectoriser packs the depthwise FMAs into v_pk_fma_f32 pairs, which
# on gfx950 run at the scalar-FMA rate but need operand pairs in adjacent registers: the LDS-DMA GroupDW
# kernel goes from 109 VGPRs to 256 + spills with it
FILE_FLAGS = {'xcorr.hip': ['-fno-slp-vectorize'],
              'conv_igemm.hip': ['-std=c++20']}      # templated lambda over the producer's register buffers


INFO = os.path.join(CSRC, 'build_info.json')      # written by build(): what the last call compiled (git-ignored, travels to the GPU box)


def csrc_tree():
    """Content hash of everything libusot_hip.so is built from (csrc/*.hip, csrc/*.h, include/*.h, this file's flags): the
    key that ties a committed counter file (profiles/pmc_*.json: `_meta.csrc_tree`) to the kernels it was measured on.  The
    GPU box has no .git, so this is a hash of file contents, not `git rev-parse HEAD:usot_amd/csrc`."""
    import hashlib
    h = hashlib.sha256()
    files = sources() + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + sorted(glob.glob(os.path.join(INCLUDE, '*.h')))
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0')
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(repr((FLAGS[:4], sorted(FILE_FLAGS.items()))).encode())     # include paths differ between boxes: the first four flags only
    if EXPERIMENTS:
        h.update(b'USOT_EXPERIMENTS')
    return h.hexdigest()[:16]


def build_info():
    """The record the last build() left (or a stub when the library was never built here)."""
    import json
    try:
        with open(INFO) as f:
            info = json.load(f)
    except Exception:
        info = {'compiled_units': None, 'note': 'no build record beside the library'}
    info['lib_exists'] = os.path.exists(LIB)
    info['lib_mtime'] = int(os.path.getmtime(LIB)) if os.path.exists(LIB) else None
    info['lib_stale'] = is_stale()
    info['csrc_tree_now'] = csrc_tree()
    return info


def _write_info(compiled, linked, seconds):
    import json, socket, time
    with open(INFO, 'w') as f:
        json.dump({'csrc_tree': csrc_tree(), 'experiments': EXPERIMENTS, 'compiled_units': compiled, 'linked': linked, 'seconds': round(seconds, 1),
                   'when': int(time.time()), 'host': socket.gethostname(), 'hipcc': _hipcc()}, f, indent=1, sort_keys=True)


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError('hipcc not found')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = (sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(INCLUDE, '*.h'))
            + [os.path.abspath(__file__)])               # FLAGS / FILE_FLAGS live in this file
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip translation unit and link the shared library.  Leaves csrc/build_info.json: which units THIS call
    compiled (an empty list = the library was up to date and nothing ran), so a driver's record can show a real build."""
    import json, time
    t0 = time.time()
    try:
        with open(INFO) as f:
            if bool(json.load(f).get('experiments', False)) != EXPERIMENTS and os.path.exists(LIB):
                force = True                       # the objects on disk were compiled with the other -DUSOT_EXPERIMENTS setting
    except Exception:
        pass
    if not force and not is_stale():
        if not os.path.exists(INFO):
            _write_info([], False, 0.0)
        return LIB
    cc = _hipcc()
    objs = []
    procs = []
    hdrs = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(INCLUDE, '*.h')) + [os.path.abspath(__file__)]
    for src in sources():
        obj = src[:-4] + '.o'
        objs.append(obj)
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in [src] + hdrs):
            continue                              # object is newer than its source and every header
        tmp = obj + '.tmp%d' % os.getpid()          # an interrupted hipcc must not leave a fresh-looking .o
        cmd = [cc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', tmp]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, obj, tmp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = None
    for src, obj, tmp, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = failed or 'hipcc failed on %s:\n%s' % (src, out.decode(errors='replace'))
            if os.path.exists(tmp):
                os.remove(tmp)
        else:
            os.replace(tmp, obj)
    if failed:
        raise RuntimeError(failed)
    tmp = LIB + '.tmp%d' % os.getpid()
    subprocess.check_call([cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs)
    os.replace(tmp, LIB)
    _write_info([os.path.basename(src) for src, _, _, _ in procs], True, time.time() - t0)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
