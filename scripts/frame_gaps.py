#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace rocpd database of the default bench: per frame (a frame starts at its stem kernel;
before round 6 at the rows_copy_multi gather in front of it), the GPU-busy time inside the frame, the gaps between its kernels, and the
gap between the frame's last kernel and the next frame's first one."""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
rows = db.execute('select s.display_name, k.start, k.end from rocpd_kernel_dispatch k '
                  'join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start').fetchall()
starts = [i for i, r in enumerate(rows) if 'stem_pool_f32' in r[0]]
if not any('rows_append_gather' in r[0] for r in rows):        # before round 6's 'defer_append' = 2 a frame began with the gather
    starts = [i - 1 for i in starts if i > 0 and 'rows_copy_multi' in rows[i - 1][0]]
frames = []
for a, b in zip(starts[:-1], starts[1:]):
    ks = rows[a:b]
    if not 36 <= len(ks) <= 70:
        continue
    busy = sum(e - s for _, s, e in ks)
    inner = sum(max(0, ks[i + 1][1] - ks[i][2]) for i in range(len(ks) - 1))
    frames.append((ks[0][1], ks[-1][2], busy, inner, rows[b][1] - ks[-1][2], rows[b][1] - ks[0][1]))
f = np.array(frames[len(frames) // 4:], float) / 1e3
print('%d frames: period median %.1f us (p10 %.1f, p90 %.1f) | kernels busy %.1f | gaps inside the frame %.1f | last kernel -> next frame\'s first %.1f (p10 %.1f, p90 %.1f)' % (
    len(f), np.median(f[:, 5]), np.percentile(f[:, 5], 10), np.percentile(f[:, 5], 90), np.median(f[:, 2]), np.median(f[:, 3]),
    np.median(f[:, 4]), np.percentile(f[:, 4], 10), np.percentile(f[:, 4], 90)))
fast = f[f[:, 5] < 900]
slow = f[f[:, 5] >= 900]
for name, g in (('period < 900 us', fast), ('period >= 900', slow)):
    if len(g):
        print('  %-14s n %4d  period %.1f  busy %.1f  inner gaps %.1f  inter-frame gap %.1f' % (name, len(g), np.median(g[:, 5]), np.median(g[:, 2]), np.median(g[:, 3]), np.median(g[:, 4])))
# where inside a slow frame the gaps sit: mean gap BEFORE kernel i (us), slow frames only
import collections
pos = collections.defaultdict(list)
names = {}
for a, b in zip(starts[:-1], starts[1:]):
    ks = rows[a:b]
    if not 36 <= len(ks) <= 70 or (rows[b][1] - ks[0][1]) / 1e3 < 900:
        continue
    for i in range(1, len(ks)):
        pos[i].append(max(0, ks[i][1] - ks[i - 1][2]) / 1e3)
        names[i] = ks[i][0][:40]
big = [(float(np.mean(v)), i) for i, v in pos.items() if np.mean(v) > 1.0]
print('slow frames: gaps > 1 us on average before kernel #: ' + ', '.join('#%d %s %.1f us' % (i, names[i], g) for g, i in sorted(big, key=lambda t: t[1])))
