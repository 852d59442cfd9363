"""CPU: counter-derived fields of the bench line are tied to the source tree they were measured on, and the line says what
the in-tree library was built from (stale `traffic` / `mfma_busy` / `clock_ghz` must be impossible)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_csrc_tree_is_a_content_hash(tmp_path, monkeypatch):
    from usot_amd import build
    a = build.csrc_tree()
    assert a == build.csrc_tree() and len(a) == 16
    # another file set -> another key
    extra = tmp_path / 'zz_extra.hip'
    extra.write_text('// not part of the library\n')
    real = build.sources
    monkeypatch.setattr(build, 'sources', lambda: real() + [str(extra)])
    assert build.csrc_tree() != a


def test_counter_files_of_another_tree_give_null_fields(monkeypatch):
    import bench
    monkeypatch.setattr(bench, '_tree_now', lambda: 'aaaaaaaaaaaaaaaa')
    ok, src = bench.counters_current({'csrc_tree': 'aaaaaaaaaaaaaaaa', 'commit': 'abc1234'}, 'profiles/x.json')
    assert ok and 'abc1234' in src
    ok, src = bench.counters_current({'csrc_tree': 'bbbbbbbbbbbbbbbb', 'commit': 'abc1234'}, 'profiles/x.json')
    assert not ok and 'STALE' in src
    ok, _ = bench.counters_current({'commit': 'old file without the field'}, 'profiles/x.json')
    assert not ok
    # the committed files against a tree that cannot be theirs: every derived field is None and the reason is carried
    f = bench.busy_fields(bench.pmc_busy(bench.LP_DOMINANT_KERNEL, bench.BF16_PEAK_TFLOPS), 800.0)
    assert f['peak_sustained'] is None and f.get('mfma_busy') is None and f.get('clock_ghz') is None
    assert 'STALE' in f.get('busy_source', 'STALE')          # (file absent: busy_fields returns the bare null form)
    t = bench.xcorr_traffic('groupdw_dma_kernel', 2048)
    assert t['traffic'] is None


def test_counter_files_in_the_tree_carry_the_key():
    """Every committed counter file names the tree it was measured on (files of earlier rounds without the key read as stale)."""
    import bench
    from usot_amd import build
    now = build.csrc_tree()
    for name in ('pmc_traffic.json', 'pmc_traffic_bf16.json', 'pmc_traffic_mixed.json', 'pmc_busy.json', 'pmc_xcorr.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            meta = json.load(f).get('_meta', {})
        ok, src = bench.counters_current(meta, 'profiles/' + name)
        assert ok == (meta.get('csrc_tree') == now), (name, src)


def test_build_info_in_the_line():
    import bench
    from usot_amd import build
    build.build()                                  # up to date: compiles nothing, leaves / keeps the record
    bi = bench.build_info_line()
    for k in ('lib', 'lib_mtime', 'lib_stale', 'csrc_tree', 'built_from_csrc_tree', 'compiled_units_last_build'):
        assert k in bi
    assert bi['lib_stale'] is False and bi['csrc_tree'] == build.csrc_tree()
    assert isinstance(bi['compiled_units_last_build'], list)
    json.dumps(bi)
