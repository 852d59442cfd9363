"""Benchmark sequence listing for the test driver (reference lib/dataset_loader/
benchmark.py:8-230): `load_dataset(name) -> {video: {'image_files', 'gt', 'name'}}` for the
single-object benchmarks scripts/test_usot.py iterates (OTB, VOT2016/18/19/20, GOT-10k
val/test, TrackingNet, LaSOT, VisDrone).  Pure host-side file discovery; datasets live in
<repo>/datasets_test/<NAME> exactly as the reference expects.  The multi-modal / VOS
layouts of the reference loader (RGBT234, VOT2019RGBT, DAVIS, YTBVOS) are not consumed
by the USOT tracker and are not listed."""
import glob
import json
import os
from os.path import dirname, join, realpath

import numpy as np

ROOT = join(realpath(dirname(__file__)), '..', 'datasets_test')


def _jpgs(*parts):
    return sorted(glob.glob(join(*parts)))


def _vot(base, gt_in_color, raw_gt):
    with open(join(base, 'list.txt')) as f:
        videos = sorted(v.strip() for v in f.readlines())
    out = {}
    for v in videos:
        files = _jpgs(base, v, '*.jpg') or _jpgs(base, v, 'color', '*.jpg')
        gt_path = join(base, v, 'color', 'groundtruth.txt') if gt_in_color else join(base, v, 'groundtruth.txt')
        if raw_gt:                       # VOT2020 masks/polygons are parsed by the caller
            with open(gt_path, 'r') as f:
                gt = f.readlines()
        else:
            gt = np.loadtxt(gt_path, delimiter=',').astype(np.float64)
        out[v] = {'image_files': files, 'gt': gt, 'name': v}
    return out


def _per_video_dirs(seq_root, gt_of, wrap=False, skip=('list.txt',), numeric_sort=False):
    out = {}
    for v in sorted(os.listdir(seq_root)):
        if v in skip or v.endswith('.json'):
            continue
        files = _jpgs(seq_root, v, '*.jpg')
        if numeric_sort:
            files.sort(key=lambda x: int(x.split('/')[-1][:-4]))
        gt = np.loadtxt(gt_of(v), delimiter=',')
        out[v] = {'image_files': files, 'gt': [gt] if wrap else gt, 'name': v}
    return out


def load_dataset(dataset):
    base = join(ROOT, dataset)
    if 'OTB' in dataset:
        with open(join(ROOT, dataset + '.json'), 'r') as f:
            info = json.load(f)
        for v in info.keys():
            info[v]['image_files'] = [join(base, f) for f in info[v]['img_names']]
            info[v]['gt'] = np.array(info[v]['gt_rect']) - [1, 1, 0, 0]
            info[v]['name'] = info[v]['video_dir']
        return info
    if 'VOT2020' in dataset:
        return _vot(base, gt_in_color=False, raw_gt=True)
    if 'VOT' in dataset and 'VOT2019RGBT' not in dataset:
        return _vot(base, gt_in_color=True, raw_gt=False)
    if 'VISDRONEVAL' in dataset:
        return _per_video_dirs(join(base, 'sequences'), lambda v: join(base, 'annotations', v + '.txt'), skip=())
    if 'VISDRONETEST' in dataset:
        out = _per_video_dirs(join(base, 'sequences'), lambda v: join(base, 'initialization', v + '.txt'), skip=())
        for v in out.values():
            v['gt'] = v['gt'].reshape(1, 4)
        return out
    if 'GOT10KVAL' in dataset:
        return _per_video_dirs(base, lambda v: join(base, v, 'groundtruth.txt'))
    if 'GOT10K' in dataset:
        return _per_video_dirs(base, lambda v: join(base, v, 'groundtruth.txt'), wrap=True)
    if 'TRACKINGNET' in dataset:
        return _per_video_dirs(join(base, 'frames'), lambda v: join(base, 'anno', v + '.txt'), wrap=True,
                               skip=(), numeric_sort=True)
    if 'LASOT' in dataset:
        with open(join(ROOT, dataset + '.json'), 'r') as f:
            wanted = set(json.load(f).keys())
        out = {}
        for v in sorted(os.listdir(base)):
            if v not in wanted:
                continue
            gt = np.loadtxt(join(base, v, 'groundtruth.txt'), delimiter=',') - [1, 1, 0, 0]
            out[v] = {'image_files': _jpgs(base, v, 'img', '*jpg'), 'gt': gt, 'name': v}
        return out
    raise ValueError("Dataset not support now, edit for other dataset youself...")
