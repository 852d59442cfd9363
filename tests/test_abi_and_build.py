"""CPU: the C-ABI library is built for gfx950 and exports every symbol include/usot_hip.h
declares; the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from usot_amd import build, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def libpath():
    return build.build(force=False)


def declared_symbols():
    with open(os.path.join(ROOT, 'include', 'usot_hip.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(usot_[a-z0-9_]+|PrRoIPooling[A-Za-z]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(libpath):
    syms = declared_symbols()
    assert len(syms) >= 30
    L = ctypes.CDLL(libpath)
    for s in syms:
        assert hasattr(L, s), s
    assert set(hip.EXPORTS) <= set(syms)
    # the reference's own native symbols (prroi_pooling_gpu_impl.cuh:20-54)
    assert {'PrRoIPoolingForwardGpu', 'PrRoIPoolingBackwardGpu', 'PrRoIPoolingCoorBackwardGpu'} <= set(syms)
    L.usot_abi_version.restype = ctypes.c_int
    assert L.usot_abi_version() == 6
    L.usot_strerror.restype = ctypes.c_char_p
    assert b'invalid' in L.usot_strerror(-1)


def test_code_object_is_gfx950(libpath):
    with open(libpath, 'rb') as f:
        blob = f.read()
    assert b'gfx950' in blob and b'gfx942' not in blob and b'sm_' not in blob


def test_no_cpu_fallback():
    from usot_amd.model import USOT
    m = USOT()
    with pytest.raises(hip.HipError):
        m.template(torch.zeros(1, 3, 127, 127))
    with pytest.raises(hip.HipError):
        hip.xcorr_depthwise(torch.zeros(1, 4, 9, 9), torch.zeros(1, 4, 3, 3))
    with pytest.raises(NotImplementedError):
        from lib.models.prroi_pool import PrRoIPool2D
        PrRoIPool2D(7, 7, 1.0)(torch.zeros(1, 4, 9, 9), torch.zeros(1, 5))
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1), torch.zeros(1))


def test_product_path_never_imports_the_oracle():
    pat = re.compile(r'^\s*(from|import)\s+(usot_oracle|oracle)\b', re.M)
    for d in ('usot_amd', 'lib'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, d)):
            for fn in files:
                if fn.endswith('.py'):
                    with open(os.path.join(dirpath, fn)) as f:
                        assert not pat.search(f.read()), os.path.join(dirpath, fn)


def test_low_precision_tuning_table_names_existing_tiles(libpath):
    """usot_amd/data/tuning_lp_gfx950.json (conv shape -> tile id of csrc/conv_bf16.hip) must only name tiles the library has:
    a stale id would make every plan of that shape fail with EINVAL on the GPU box."""
    import ctypes
    import json
    import os
    n = ctypes.CDLL(libpath).usot_conv_bf16_tile_count()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(here, 'usot_amd', 'data', 'tuning_lp_gfx950.json')) as f:
        table = json.load(f)
    assert n >= 37 and table
    for key, tile in table.items():
        m, cout, k = (int(v) for v in key.split(','))
        assert m > 0 and cout > 0 and k % 64 == 0, key
        assert 1 <= tile <= n, (key, tile, n)


def test_split16_tuning_table_names_split_fp16_tiles(libpath):
    """usot_amd/data/tuning_split16_gfx950.json (fp32 conv shape -> (tile, ksplit) on the split-fp16 tiles) and engine.SPLIT16_TILES
    (fp32 tile -> its split-fp16 twin) must only name tiles whose filters the library takes pre-split (usot_conv_tile_wfrag == 2),
    of the twin's own shape; the reductions they serve are whole 64-k tiles."""
    import ctypes
    import json
    L = ctypes.CDLL(libpath)
    with open(os.path.join(ROOT, 'usot_amd', 'data', 'tuning_split16_gfx950.json')) as f:
        table = json.load(f)
    assert table
    for key, (tile, ks) in table.items():
        m, cout, k, groups = (int(v) for v in key.split(','))
        assert m > 0 and cout > 0 and groups >= 1 and k % 64 == 0 and ks >= 1, key
        assert 1 <= tile <= L.usot_conv_tile_count() and L.usot_conv_tile_wfrag(tile) == 2, (key, tile)
    from usot_amd import engine
    bm, bn, bm2, bn2 = (ctypes.c_int() for _ in range(4))
    for src, dst in engine.SPLIT16_TILES.items():
        assert L.usot_conv_tile_wfrag(src) == 0 and L.usot_conv_tile_wfrag(dst) == 2, (src, dst)
        assert L.usot_conv_tile_info(src, ctypes.byref(bm), ctypes.byref(bn)) == 0
        assert L.usot_conv_tile_info(dst, ctypes.byref(bm2), ctypes.byref(bn2)) == 0
        assert (bm.value, bn.value) == (bm2.value, bn2.value), (src, dst)


def test_fused_bottleneck_shape_queries_are_host_functions(libpath):
    """The shape / panel queries of the fused bottleneck kernels (csrc/conv_pw_lp.hip) answer without a GPU: layer3 and layer2 widths,
    the pair forms, and the panel rule (128-pixel panels below 192 panels of 256 and for a mostly empty second round)."""
    L = ctypes.CDLL(libpath)
    L.usot_conv_pw_pixels.argtypes = [ctypes.c_int64]
    assert L.usot_conv_pw_supported(256, 256, 1024) == 1 and L.usot_conv_pw_supported(128, 128, 512) == 1
    assert L.usot_conv_pw_supported(64, 64, 256) == 0 and L.usot_conv_pw_supported(256, 256, 512) == 0
    assert L.usot_conv_pw_pair_supported(128, 512, 128) == 1 and L.usot_conv_pw_pair_supported(128, 512, 256) == 1
    assert L.usot_conv_pw_pair_supported(256, 1024, 256) == 1 and L.usot_conv_pw_pair_supported(256, 1024, 128) == 0
    assert L.usot_conv_pw_pixels(32 * 961) == 128           # batch 32 at layer3 resolution: 121 panels of 256
    assert L.usot_conv_pw_pixels(64 * 961) == 256           # batch 64: 241 panels, one round
    assert L.usot_conv_pw_pixels(64 * 1089) == 128          # 271 x 271 crops: 273 panels of 256 would leave a second round 7 % full
    assert L.usot_conv_pw_pixels(192 * 961) == 256          # many rounds


def test_default_library_holds_exactly_the_routed_conv_tiles(libpath):
    """the product library compiles the conv tiles something can SELECT - the tuning tables
    (usot_amd/data/tuning_gfx950.json, tuning_split16_gfx950.json, tuning_lp_gfx950.json), engine.SPLIT16_TILES' twins of tuned tiles,
    the engine's deferred-launch options and default batch tile, and the launchers' own heuristics - and nothing else; every other
    tile id of the tables is an experiment that exists only in a USOT_EXPERIMENTS=1 build (parity tests: `-m experiments`)."""
    import json
    from usot_amd import engine
    L = ctypes.CDLL(libpath)
    if L.usot_experiments_built():
        pytest.skip('experiments build: every tile id is compiled')
    data = os.path.join(ROOT, 'usot_amd', 'data')
    with open(os.path.join(data, 'tuning_gfx950.json')) as f:
        f32 = {int(v[0]) for k, v in json.load(f).items() if not k.startswith('_')}
    with open(os.path.join(data, 'tuning_split16_gfx950.json')) as f:
        s16 = {int(v[0]) for v in json.load(f).values()}
    with open(os.path.join(data, 'tuning_lp_gfx950.json')) as f:
        lp = {int(v) for v in json.load(f).values()}
    opt = engine.DEFAULT_OPTIONS
    routed = set(f32) | s16 | {1, 2, 4, 5, 7, 8}                     # pick_tile() in csrc/conv_igemm.hip
    routed |= {15}                                                   # Builder.default_batch_tile
    routed |= {engine.SPLIT16_TILES[t] for t in f32 if t in engine.SPLIT16_TILES}
    for key in ('defer_split_f32', 'defer_split_s16', 'defer_split_res_f32', 'batch_ds_conv2'):
        routed |= {int(v[0]) for v in opt[key].values()}
    built = {t for t in range(1, L.usot_conv_tile_count() + 1) if L.usot_conv_tile_built(t)}
    assert built == routed, (sorted(built - routed), sorted(routed - built))
    lp_routed = lp | {1, 4, 5} | {21}                                # heuristic of usot_conv2d_lp; 21 = the plain form of tile 32's loop
    lp_built = {t for t in range(1, L.usot_conv_bf16_tile_count() + 1) if L.usot_conv_bf16_tile_built(t)}
    assert lp_built == lp_routed, (sorted(lp_built - lp_routed), sorted(lp_routed - lp_built))
    assert os.path.getsize(libpath) < 4 * 1024 * 1024                # 8.5 MB with the experiments
