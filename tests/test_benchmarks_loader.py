"""CPU: usot_amd.benchmarks.load_dataset against the REFERENCE's loader
(lib/dataset_loader/benchmark.py:8-230) — fixture tests/golden/golden_datasets.json was produced by
running the reference's own function on the fake datasets_test/ tree of tests/golden/fake_datasets.py
(make_golden.py datasets); here the same tree is rebuilt and listed by this repo's loader.  Every
layout, key, value, dtype and the video ORDER must agree."""
import json
import os

import numpy as np
import pytest

import fake_datasets
from usot_amd import benchmarks

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_datasets.json')


@pytest.fixture(scope='module')
def tree(tmp_path_factory):
    root = os.path.realpath(str(tmp_path_factory.mktemp('fake')))
    fake_datasets.build(root)
    old = benchmarks.ROOT
    benchmarks.ROOT = os.path.join(root, 'datasets_test')
    yield root
    benchmarks.ROOT = old


@pytest.fixture(scope='module')
def gold():
    with open(GOLD) as f:
        return json.load(f)


@pytest.mark.parametrize('name', fake_datasets.DATASETS)
def test_load_dataset_matches_reference(tree, gold, name):
    got = fake_datasets.normalise(benchmarks.load_dataset(name), tree)
    want = gold[name]
    assert got['order'] == want['order']
    assert json.loads(json.dumps(got['videos'], sort_keys=True)) == want['videos']


def test_every_reference_layout_is_covered(gold):
    assert set(fake_datasets.DATASETS) == {k for k in gold if not k.startswith('__')}
    # and the interesting cases really are in the fixture
    assert [os.path.basename(p) for p in gold['TRACKINGNET']['videos']['0-6LB4FqxoE_0']['image_files']] == \
        ['0.jpg', '1.jpg', '2.jpg', '9.jpg', '10.jpg', '11.jpg']                     # numeric, not lexicographic
    assert 'basketball-3' not in gold['LASOTTEST']['order']                         # only the videos the json names
    assert gold['DAVIS2017']['order'] == ['bike-packing', 'blackswan']              # val.txt order
    assert gold['VOT2020']['videos']['agility']['gt'][0].startswith('m10,')         # raw lines, parsed by the caller
    assert gold['GOT10KTEST']['order'] == ['GOT-10k_X_000001', 'GOT-10k_X_000002']  # list.txt and meta.json skipped


def test_unsupported_and_malformed(tree, gold):
    with pytest.raises(ValueError) as e:
        benchmarks.load_dataset('NOSUCH')
    assert str(e.value) == gold['__unsupported__']
    # GOT-10k without its list.txt: the reference's `videos.remove('list.txt')` raises ValueError
    os.rename(os.path.join(tree, 'datasets_test', 'GOT10KVAL', 'list.txt'), os.path.join(tree, 'datasets_test', 'GOT10KVAL.list'))
    try:
        with pytest.raises(ValueError):
            benchmarks.load_dataset('GOT10KVAL')
    finally:
        os.rename(os.path.join(tree, 'datasets_test', 'GOT10KVAL.list'), os.path.join(tree, 'datasets_test', 'GOT10KVAL', 'list.txt'))


def test_gt_dtypes_and_shapes(tree):
    otb = benchmarks.load_dataset('OTB2015')
    assert otb['Basketball']['gt'].shape == (3, 4) and otb['Jogging.2']['name'] == 'Jogging'
    vis = benchmarks.load_dataset('VISDRONETEST')
    assert all(v['gt'].shape == (1, 4) for v in vis.values())
    got = benchmarks.load_dataset('GOT10KTEST')
    assert all(isinstance(v['gt'], list) and v['gt'][0].shape == (4,) for v in got.values())
    vot = benchmarks.load_dataset('VOT2018')
    assert all(v['gt'].dtype == np.float64 and v['gt'].shape == (3, 8) for v in vot.values())
    assert all('/color/' in f for v in vot.values() for f in v['image_files'])
    y = benchmarks.load_dataset('YTBVOS')['0062f687f1']
    assert y['start_frame'] == {'1': 0, '2': 1} and y['end_frame'] == {'1': 2, '2': 3}
