#!/usr/bin/env python3
"""Per-kernel mean of one PMC counter (e.g. FETCH_SIZE / WRITE_SIZE, KB per dispatch) from a
rocprofv3 --pmc rocpd database."""
import re, sqlite3, sys


def short(name):
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*$', '', name)[:90]


def main(path, out=None):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(rocpd_kernel_dispatch)')]
    key = 'event_id' if 'event_id' in cols else 'id'
    rows = db.execute(
        'select s.display_name, i.name, p.value from rocpd_pmc_event p '
        'join rocpd_info_pmc i on p.pmc_id = i.id '
        'join rocpd_kernel_dispatch k on k.%s = p.event_id '
        'join rocpd_info_kernel_symbol s on k.kernel_id = s.id' % key).fetchall()
    agg = {}
    for name, ctr, v in rows:
        a = agg.setdefault((short(name), ctr), [0, 0.0])
        a[0] += 1
        a[1] += v
    lines = ['%-92s %-12s %8s %14s %14s' % ('kernel', 'counter', 'calls', 'mean', 'total')]
    for (name, ctr), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('%-92s %-12s %8d %14.2f %14.1f' % (name, ctr, n, tot / n, tot))
    text = '\n'.join(lines)
    if out:
        open(out, 'w').write(text + '\n')
    print(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
