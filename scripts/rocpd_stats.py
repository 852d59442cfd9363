#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into the per-kernel table
`rocprofv3 --stats` prints: calls, total / average / min / max duration, share of GPU time."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'\((anonymous namespace::)?[A-Za-z]+K( const)?\)$', '', name)
    return name[:110]


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        'select s.display_name, k.start, k.end from rocpd_kernel_dispatch k '
        'join rocpd_info_kernel_symbol s on k.kernel_id = s.id').fetchall()
    agg = {}
    for name, st, en in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = en - st
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    lines = ['%-112s %8s %14s %12s %10s %10s %7s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'pct')]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('%-112s %8d %14d %12.1f %10d %10d %6.2f%%' % (name, a[0], a[1], a[1] / a[0], a[2], a[3], 100.0 * a[1] / tot))
    lines.append('TOTAL kernel time %.3f ms over %d dispatches' % (tot / 1e6, len(rows)))
    text = '\n'.join(lines)
    if out:
        with open(out, 'w') as f:
            f.write(text + '\n')
    print(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
