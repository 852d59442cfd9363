"""A small fake `datasets_test/` tree covering every layout the reference's load_dataset reads
(lib/dataset_loader/benchmark.py:8-230): built identically by the fixture generator (which points the
REFERENCE loader at it) and by the CPU test (which points this repo's loader at it).  Images are empty
files: the loader only lists them."""
import json
import os

import numpy as np


def _touch(path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, 'wb').close()


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        f.write(text)


def _gt_rows(seed, n, cols=4):
    g = np.random.default_rng(seed)
    return np.round(g.uniform(1, 200, (n, cols)), 2)


def _csv(a):
    return '\n'.join(','.join('%.2f' % v for v in row) for row in np.atleast_2d(a)) + '\n'


DATASETS = ('OTB2015', 'VOT2016', 'VOT2018', 'VOT2020', 'RGBT234', 'VOT2019RGBT', 'VISDRONEVAL', 'VISDRONETEST',
            'GOT10KVAL', 'GOT10KTEST', 'TRACKINGNET', 'LASOTTEST', 'DAVIS2017', 'YTBVOS')


def build(root):
    """Create <root>/datasets_test/... ; returns the datasets_test path."""
    d = os.path.join(root, 'datasets_test')
    # OTB: a json index, 1-based rects
    otb = {}
    for i, v in enumerate(('Basketball', 'Jogging.2')):
        names = ['%s/img/%04d.jpg' % (v.split('.')[0], k + 1) for k in range(3)]
        for n in names:
            _touch(os.path.join(d, 'OTB2015', n))
        otb[v] = {'video_dir': v.split('.')[0], 'img_names': names, 'init_rect': _gt_rows(i, 1)[0].tolist(),
                  'gt_rect': _gt_rows(10 + i, 3).tolist()}
    _write(os.path.join(d, 'OTB2015.json'), json.dumps(otb))
    # VOT2016: frames beside the video; VOT2018: frames under color/ ; both read color/groundtruth.txt (8 numbers)
    for ds, in_color in (('VOT2016', False), ('VOT2018', True)):
        vids = ['zebra', 'ants1', 'bag']
        _write(os.path.join(d, ds, 'list.txt'), '\n'.join(vids) + '\n')
        for i, v in enumerate(vids):
            for k in (3, 1, 2):
                _touch(os.path.join(d, ds, v, 'color' if in_color else '', '%08d.jpg' % k))
            _write(os.path.join(d, ds, v, 'color', 'groundtruth.txt'), _csv(_gt_rows(20 + i, 3, 8)))
    # VOT2020: raw ground-truth lines (masks / rects), groundtruth.txt beside color/
    vids = ['agility', 'ball3']
    _write(os.path.join(d, 'VOT2020', 'list.txt'), '\n'.join(vids) + '\n')
    for i, v in enumerate(vids):
        for k in (1, 2):
            _touch(os.path.join(d, 'VOT2020', v, 'color', '%08d.jpg' % k))
        _write(os.path.join(d, 'VOT2020', v, 'groundtruth.txt'), 'm10,20,30,40,1,2,3\n5.0,6.0,70.5,80.25\n')
    # RGBT234: json index with separate infrared / visible lists
    rg = {}
    for i, v in enumerate(('afterrain', 'bike')):
        rg[v] = {'name': v, 'infrared_imgs': ['%05di.jpg' % k for k in range(2)], 'visiable_imgs': ['%05dv.jpg' % k for k in range(2)],
                 'infrared_gt': _gt_rows(30 + i, 2).tolist(), 'visiable_gt': _gt_rows(40 + i, 2).tolist()}
    _write(os.path.join(d, 'RGBT234.json'), json.dumps(rg))
    # VOT2019RGBT: ir/ and color/ folders
    vids = ['car10', 'biketwo']
    _write(os.path.join(d, 'VOT2019RGBT', 'list.txt'), '\n'.join(vids) + '\n')
    for i, v in enumerate(vids):
        for k in (2, 1):
            _touch(os.path.join(d, 'VOT2019RGBT', v, 'ir', '%05di.jpg' % k))
            _touch(os.path.join(d, 'VOT2019RGBT', v, 'color', '%05dv.jpg' % k))
        _write(os.path.join(d, 'VOT2019RGBT', v, 'groundtruth.txt'), _csv(_gt_rows(50 + i, 2, 8)))
    # VisDrone val / test
    for ds, anno, rows in (('VISDRONEVAL', 'annotations', 3), ('VISDRONETEST', 'initialization', 1)):
        for i, v in enumerate(('uav0000086_00000_s', 'uav0000117_02622_s')):
            for k in range(1, 4):
                _touch(os.path.join(d, ds, 'sequences', v, 'img%07d.jpg' % k))
            _write(os.path.join(d, ds, anno, v + '.txt'), _csv(_gt_rows(60 + i, rows)))
    # GOT-10k val / test: list.txt beside the video folders (test also holds a json)
    for ds, rows in (('GOT10KVAL', 3), ('GOT10KTEST', 1)):
        vids = ['GOT-10k_X_000002', 'GOT-10k_X_000001']
        _write(os.path.join(d, ds, 'list.txt'), '\n'.join(vids) + '\n')
        if ds == 'GOT10KTEST':
            _write(os.path.join(d, ds, 'meta.json'), '{}')
        for i, v in enumerate(vids):
            for k in range(1, 4):
                _touch(os.path.join(d, ds, v, '%08d.jpg' % k))
            _write(os.path.join(d, ds, v, 'groundtruth.txt'), _csv(_gt_rows(70 + i, rows)))
    # TrackingNet: frames/<video>/<n>.jpg with UNPADDED numbers (numeric sort), anno/<video>.txt
    for i, v in enumerate(('0-6LB4FqxoE_0', 'zz1hcZ8YaDk_0')):
        for k in (0, 1, 2, 10, 11, 9):
            _touch(os.path.join(d, 'TRACKINGNET', 'frames', v, '%d.jpg' % k))
        _write(os.path.join(d, 'TRACKINGNET', 'anno', v + '.txt'), _csv(_gt_rows(80 + i, 1)))
    _write(os.path.join(d, 'TRACKINGNET', 'frames', 'index.json'), '{}')
    # LaSOT: json names the test videos; folders hold more
    _write(os.path.join(d, 'LASOTTEST.json'), json.dumps({'airplane-1': {}, 'zebra-17': {}}))
    for i, v in enumerate(('airplane-1', 'basketball-3', 'zebra-17')):
        for k in range(1, 4):
            _touch(os.path.join(d, 'LASOTTEST', v, 'img', '%08d.jpg' % k))
        _write(os.path.join(d, 'LASOTTEST', v, 'groundtruth.txt'), _csv(_gt_rows(90 + i, 3)))
    # DAVIS 2017 val
    vids = ['bike-packing', 'blackswan']
    _write(os.path.join(d, 'DAVIS', 'ImageSets', '2017', 'val.txt'), '\n'.join(vids) + '\n')
    for v in vids:
        for k in (2, 0, 1):
            _touch(os.path.join(d, 'DAVIS', 'JPEGImages', '480p', v, '%05d.jpg' % k))
            _touch(os.path.join(d, 'DAVIS', 'Annotations', '480p', v, '%05d.png' % k))
    # YouTube-VOS valid
    meta = {'videos': {'0062f687f1': {'objects': {'1': {'frames': ['00000', '00005', '00010']},
                                                  '2': {'frames': ['00005', '00010', '00015']}}},
                       '01c88b5b60': {'objects': {'1': {'frames': ['00020', '00025']}}}}}
    _write(os.path.join(d, 'YTBVOS', 'valid', 'meta.json'), json.dumps(meta))
    return d


def normalise(info, root):
    """JSON-able form of a load_dataset result with paths relative to `root` (key order kept)."""
    def conv(v):
        if isinstance(v, np.ndarray):
            return {'ndarray': v.tolist(), 'dtype': str(v.dtype)}
        if isinstance(v, (np.floating, np.integer)):
            return v.item()
        if isinstance(v, str):
            v = os.path.normpath(v) if os.sep in v and not v.endswith('\n') else v
            return os.path.relpath(v, root) if os.path.isabs(v) else v
        if isinstance(v, (list, tuple)):
            return [conv(e) for e in v]
        if isinstance(v, dict):
            return {str(k): conv(e) for k, e in v.items()}
        return v
    return {'order': [str(k) for k in info.keys()], 'videos': conv(dict(info))}
