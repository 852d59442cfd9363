"""Deterministic sub-sampling of big tensors so golden fixtures stay small."""
import zlib

import numpy as np

N_SAMPLES = 2048


def sample_index(name, numel, n=N_SAMPLES):
    g = np.random.default_rng(zlib.crc32(name.encode()))
    return g.integers(0, numel, min(n, numel))


def sample_index2(name, numel, n=N_SAMPLES):
    """A second sample from another seed, DISJOINT from sample_index(name, numel): an error confined to elements the first
    2 048 points miss has a second, independent chance of being seen."""
    first = set(sample_index(name, numel, n).tolist())
    g = np.random.default_rng(zlib.crc32((name + '#second').encode()) ^ 0x5bd1e995)
    out = []
    while len(out) < min(n, max(0, numel - len(first))):
        for v in g.integers(0, numel, n).tolist():
            if v not in first:
                first.add(v)
                out.append(v)
                if len(out) == n:
                    break
    return np.array(out[:n], np.int64)


def summarize(name, arr, full_below=8192):
    """{name+'/full'} for small tensors, else {'/samp', '/stats'} (mean, |mean|, std, max|.|)."""
    a = np.asarray(arr, dtype=np.float32)
    out = {name + '/shape': np.array(a.shape, np.int64)}
    if a.size <= full_below:
        out[name + '/full'] = a
    else:
        f = a.reshape(-1)
        out[name + '/samp'] = f[sample_index(name, f.size)]
        out[name + '/samp2'] = f[sample_index2(name, f.size)]
        out[name + '/stats'] = np.array([f.mean(dtype=np.float64), np.abs(f).mean(dtype=np.float64),
                                         f.std(dtype=np.float64), np.abs(f).max()], np.float64)
    return out


def check(name, gold, arr, rtol, atol_scale=1.0):
    """Compare `arr` with a fixture written by summarize(); returns max scaled error."""
    a = np.asarray(arr, dtype=np.float32)
    assert tuple(gold[name + '/shape']) == a.shape, (name, gold[name + '/shape'], a.shape)
    if name + '/full' in gold:
        ref, got = gold[name + '/full'], a
    else:
        ref, got = gold[name + '/samp'], a.reshape(-1)[sample_index(name, a.size)]
        st = gold[name + '/stats']                      # mean, mean|.|, std, max|.| of the WHOLE tensor (float64)
        f = a.reshape(-1)
        assert abs(f.mean(dtype=np.float64) - st[0]) <= rtol * (abs(st[1]) + 1e-12) * 4, (name, 'mean')
        # whole-tensor second moment and extreme value: an error confined to elements the 2 048 sample points miss (one bad
        # tile, one bad channel) moves these.  Element-wise errors <= rtol x max(|ref|, mean|ref|) bound the change of the std by
        # rtol x sqrt(std^2 + 2 mean|.|^2) (triangle inequality on the centred vectors) and of the maximum by rtol x max|.|.
        assert abs(f.std(dtype=np.float64) - st[2]) <= rtol * np.sqrt(st[2] ** 2 + 2 * st[1] ** 2) * 2 + 1e-12, \
            (name, 'std', float(f.std(dtype=np.float64)), float(st[2]))
        assert abs(float(np.abs(f).max()) - st[3]) <= rtol * st[3] * 2 + 1e-12, (name, 'max|.|', float(np.abs(f).max()), float(st[3]))
    scale = np.maximum(np.abs(ref), atol_scale * np.abs(ref).mean() + 1e-30)
    err = float(np.max(np.abs(got - ref) / scale))
    assert err <= rtol, '%s: scaled error %.3e > %.1e' % (name, err, rtol)
    if name + '/samp2' in gold:                         # the second, disjoint sample (fixtures written from round 5 on)
        ref2, got2 = gold[name + '/samp2'], a.reshape(-1)[sample_index2(name, a.size)]
        scale2 = np.maximum(np.abs(ref2), atol_scale * np.abs(ref).mean() + 1e-30)
        err2 = float(np.max(np.abs(got2 - ref2) / scale2))
        assert err2 <= rtol, '%s (second sample): scaled error %.3e > %.1e' % (name, err2, rtol)
        err = max(err, err2)
    return err
