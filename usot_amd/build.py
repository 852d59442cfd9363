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
# per-file flags.  xcorr.hip: hipcc's SLP vectoriser packs the depthwise FMAs into v_pk_fma_f32 pairs, which
# on gfx950 run at the scalar-FMA rate but need operand pairs in adjacent registers: the LDS-DMA GroupDW
# kernel goes from 109 VGPRs to 256 + spills with it
FILE_FLAGS = {'xcorr.hip': ['-fno-slp-vectorize'],
              'conv_igemm.hip': ['-std=c++20']}      # templated lambda over the producer's register buffers


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
    """Compile every .hip translation unit and link the shared library."""
    if not force and not is_stale():
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
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
