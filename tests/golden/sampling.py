"""Deterministic sub-sampling of big tensors so golden fixtures stay small."""
import zlib

import numpy as np

N_SAMPLES = 2048


def sample_index(name, numel, n=N_SAMPLES):
    g = np.random.default_rng(zlib.crc32(name.encode()))
    return g.integers(0, numel, min(n, numel))


def summarize(name, arr, full_below=8192):
    """{name+'/full'} for small tensors, else {'/samp', '/stats'} (mean, |mean|, std, max|.|)."""
    a = np.asarray(arr, dtype=np.float32)
    out = {name + '/shape': np.array(a.shape, np.int64)}
    if a.size <= full_below:
        out[name + '/full'] = a
    else:
        f = a.reshape(-1)
        out[name + '/samp'] = f[sample_index(name, f.size)]
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
        st = gold[name + '/stats']
        f = a.reshape(-1)
        assert abs(f.mean(dtype=np.float64) - st[0]) <= rtol * (abs(st[1]) + 1e-12) * 4, (name, 'mean')
    scale = np.maximum(np.abs(ref), atol_scale * np.abs(ref).mean() + 1e-30)
    err = float(np.max(np.abs(got - ref) / scale))
    assert err <= rtol, '%s: scaled error %.3e > %.1e' % (name, err, rtol)
    return err
