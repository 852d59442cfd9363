"""Benchmark sequence listing for the test drivers (reference lib/dataset_loader/benchmark.py:8-230):
`load_dataset(name) -> {video: {...}}` for every layout the reference reads — OTB (json index),
VOT2016/18/19 and VOT2020, RGBT234, VOT2019RGBT, VisDrone val/test, GOT-10k val/test, TrackingNet,
LaSOT, DAVIS and YouTube-VOS.  Pure host-side file discovery; datasets live in
`<repo>/datasets_test/<NAME>` exactly as the reference expects (it resolves `../../datasets_test`
from its own file).

Pinned by tests/golden/golden_datasets.json: the reference's loader run on a fake tree
(tests/golden/make_golden.py datasets), compared key for key, order included.  The name matching is
by SUBSTRING and order-sensitive in the reference ('VOT2019RGBT' contains 'VOT', 'GOT10KVAL'
contains 'GOT10K'); `_RULES` keeps that order.
"""
import glob
import json
import os
from os.path import dirname, join, realpath

import numpy as np

ROOT = join(realpath(dirname(__file__)), '..', 'datasets_test')


def _listed(base):
    """Video names of a VOT-style list.txt, sorted (benchmark.py:29-32)."""
    with open(join(base, 'list.txt')) as f:
        return sorted(line.strip() for line in f.readlines())


def _frames(*pattern):
    return sorted(glob.glob(join(*pattern)))


def _boxes(path):
    return np.loadtxt(path, delimiter=',')


def _otb(ds):
    """json index {video: {video_dir, img_names, gt_rect (1-based x, y)}} (benchmark.py:15-23)."""
    base = join(ROOT, ds)
    with open(join(ROOT, ds + '.json'), 'r') as f:
        info = json.load(f)
    for v in info.values():
        v['image_files'] = [join(base, name) for name in v['img_names']]
        v['gt'] = np.array(v['gt_rect']) - [1, 1, 0, 0]
        v['name'] = v['video_dir']
    return info


def _vot(ds, raw_gt):
    """list.txt + <video>/[color/]*.jpg.  VOT2016-19 read color/groundtruth.txt as float64 polygons
    (:25-40); VOT2020 keeps the raw lines of <video>/groundtruth.txt for the caller to parse (:42-57)."""
    base = join(ROOT, ds)
    out = {}
    for v in _listed(base):
        files = _frames(base, v, '*.jpg') or _frames(base, v, 'color', '*.jpg')
        if raw_gt:
            with open(join(base, v, 'groundtruth.txt'), 'r') as f:
                gt = f.readlines()
        else:
            gt = _boxes(join(base, v, 'color', 'groundtruth.txt')).astype(np.float64)
        out[v] = {'image_files': files, 'gt': gt, 'name': v}
    return out


def _rgbt234(ds):
    """json index with per-modality frame lists and 0-based boxes (:59-72)."""
    base = join(ROOT, ds)
    with open(join(ROOT, ds + '.json'), 'r') as f:
        info = json.load(f)
    for key, v in info.items():
        folder = v['name']
        v['infrared_imgs'] = [join(base, folder, 'infrared', n) for n in v['infrared_imgs']]
        v['visiable_imgs'] = [join(base, folder, 'visible', n) for n in v['visiable_imgs']]
        v['infrared_gt'] = np.array(v['infrared_gt'])
        v['visiable_gt'] = np.array(v['visiable_gt'])
        v['name'] = key
    return info


def _vot_rgbt(ds):
    """list.txt + <video>/{ir,color}/*.jpg + <video>/groundtruth.txt (:74-89)."""
    base = join(ROOT, ds)
    out = {}
    for v in _listed(base):
        ir, rgb = _frames(base, v, 'ir', '*.jpg'), _frames(base, v, 'color', '*.jpg')
        assert len(ir) > 0, 'please check RGBT-VOT dataloader'
        out[v] = {'infrared_imgs': ir, 'visiable_imgs': rgb,
                  'gt': _boxes(join(base, v, 'groundtruth.txt')).astype(np.float64), 'name': v}
    return out


def _visdrone(ds, anno_dir, one_box):
    """sequences/<video>/*.jpg + <anno_dir>/<video>.txt (:91-121); the test split has the first box only."""
    base = join(ROOT, ds)
    out = {}
    for v in sorted(os.listdir(join(base, 'sequences'))):
        gt = _boxes(join(base, anno_dir, v + '.txt'))
        out[v] = {'image_files': _frames(base, 'sequences', v, '*.jpg'), 'gt': gt.reshape(1, 4) if one_box else gt, 'name': v}
    return out


def _got10k(ds, test_split):
    """<video>/*.jpg + <video>/groundtruth.txt beside a list.txt, which must exist (`videos.remove`,
    :127,142); the test split skips entries with 'json' in the name and wraps the single box in a
    list (:123-151)."""
    base = join(ROOT, ds)
    videos = sorted(os.listdir(base))
    videos.remove('list.txt')
    out = {}
    for v in videos:
        if test_split and 'json' in v:
            continue
        gt = _boxes(join(base, v, 'groundtruth.txt'))
        out[v] = {'image_files': _frames(base, v, '*.jpg'), 'gt': [gt] if test_split else gt, 'name': v}
    return out


def _trackingnet(ds):
    """frames/<video>/<n>.jpg with unpadded numbers (numeric order) + anno/<video>.txt (:153-166)."""
    frames = join(ROOT, ds, 'frames')
    out = {}
    for v in sorted(os.listdir(frames)):
        if v.endswith('.json'):
            continue
        files = _frames(frames, v, '*.jpg')
        files.sort(key=lambda path: int(path.split('/')[-1][:-4]))
        out[v] = {'image_files': files, 'gt': [_boxes(join(frames, '..', 'anno', v + '.txt'))], 'name': v}
    return out


def _lasot(ds):
    """Folders named in <ds>.json only; <video>/img/*jpg; 1-based boxes (:168-186)."""
    base = join(ROOT, ds)
    with open(join(ROOT, ds + '.json'), 'r') as f:
        wanted = list(json.load(f).keys())
    out = {}
    for v in sorted(os.listdir(base)):
        if v in wanted:
            out[v] = {'image_files': _frames(base, v, 'img', '*jpg'),
                      'gt': _boxes(join(base, v, 'groundtruth.txt')) - [1, 1, 0, 0], 'name': v}
    return out


def _davis(ds):
    """DAVIS/ImageSets/<year>/val.txt order (not sorted); 480p frames and masks (:188-198)."""
    base = join(ROOT, 'DAVIS')
    with open(join(base, 'ImageSets', ds[-4:], 'val.txt')) as f:
        videos = [line.strip() for line in f.readlines()]
    return {v: {'anno_files': _frames(base, 'Annotations/480p', v, '*.png'),
                'image_files': _frames(base, 'JPEGImages/480p', v, '*.jpg'), 'name': v} for v in videos}


def _ytbvos(ds):
    """YTBVOS/valid/meta.json: the union of every object's frames, the first frame of each object as its
    initial mask, and per-object first / last positions in that union (:200-225)."""
    base = join(ROOT, 'YTBVOS', 'valid')
    with open(join(base, 'meta.json'), 'r') as f:
        videos = json.load(f)['videos']
    out = {}
    for v, meta in videos.items():
        objects = meta['objects']
        frames = sorted(np.unique([fr for obj in objects for fr in objects[obj]['frames']]))
        out[v] = {
            'anno_files': [join(base, 'Annotations', v, fr + '.png') for fr in frames],
            'anno_init_files': [join(base, 'Annotations', v, objects[obj]['frames'][0] + '.png') for obj in objects],
            'image_files': [join(base, 'JPEGImages', v, fr + '.jpg') for fr in frames],
            'name': v,
            'start_frame': {obj: frames.index(objects[obj]['frames'][0]) for obj in objects},
            'end_frame': {obj: frames.index(objects[obj]['frames'][-1]) for obj in objects},
        }
    return out


# (predicate on the dataset name, reader) in the reference's if/elif order
_RULES = (
    (lambda d: 'OTB' in d, _otb),
    (lambda d: 'VOT' in d and 'VOT2019RGBT' not in d and 'VOT2020' not in d, lambda d: _vot(d, raw_gt=False)),
    (lambda d: 'VOT2020' in d, lambda d: _vot(d, raw_gt=True)),
    (lambda d: 'RGBT234' in d, _rgbt234),
    (lambda d: 'VOT2019RGBT' in d, _vot_rgbt),
    (lambda d: 'VISDRONEVAL' in d, lambda d: _visdrone(d, 'annotations', one_box=False)),
    (lambda d: 'VISDRONETEST' in d, lambda d: _visdrone(d, 'initialization', one_box=True)),
    (lambda d: 'GOT10KVAL' in d, lambda d: _got10k(d, test_split=False)),
    (lambda d: 'GOT10K' in d, lambda d: _got10k(d, test_split=True)),
    (lambda d: 'TRACKINGNET' in d, _trackingnet),
    (lambda d: 'LASOT' in d, _lasot),
    (lambda d: 'DAVIS' in d and 'TEST' not in d, _davis),
    (lambda d: 'YTBVOS' in d, _ytbvos),
)


def load_dataset(dataset):
    for matches, reader in _RULES:
        if matches(dataset):
            return reader(dataset)
    raise ValueError("Dataset not support now, edit for other dataset youself...")
