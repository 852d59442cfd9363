#!/usr/bin/env python3
"""HBM traffic of the batch-64 low-precision backbone step (BASELINE configs[2]) from two rocprofv3 --pmc passes of
`bench.py --workload backbone_bf16` (FETCH_SIZE, WRITE_SIZE: separate runs, the TCC block has 4 counter slots), with the gfx950
correction of MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests at 64 bytes for wide coalesced reads -> doubled.

The kernels of a step are taken from the DISPATCH ORDER, not from their names: the trace is cut at every launch of the step's
first kernel (`stem_pool_lp*`), the most frequent kernel sequence between two cuts IS the step (graph replays and the eager
passes of plan.profile all produce it; warm-up segments that also hold torch copy kernels are dropped), and every dispatch of
those segments is summed.  (Round 3 selected by name prefix and silently dropped the kernels added that round:
pw_kstream_kernel, conv3x3_halo_kernel -> a traffic figure BELOW the algorithmic bytes.)

Writes profiles/pmc_traffic_bf16.json: bytes per STEP, per launch of every kernel symbol, and the step's launch list;
bench.py reads it for `backbone_bf16_b64.roofline.traffic`.
    python scripts/pmc_lp_traffic.py FETCH.db WRITE.db out.json [commit] [head-kernel-prefix] [command]"""
import collections, json, re, sqlite3, sys

def _csrc_tree():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from usot_amd import build
    return build.csrc_tree()



def short(name):
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*$', '', name)[:80]


def dispatches(path):
    """[(kernel short name, counter value)] in dispatch order."""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(rocpd_kernel_dispatch)')]
    key = 'event_id' if 'event_id' in cols else 'id'
    order = 'k.start' if 'start' in cols else 'k.id'
    rows = db.execute('select s.display_name, p.value from rocpd_pmc_event p '
                      'join rocpd_kernel_dispatch k on k.%s = p.event_id '
                      'join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by %s, k.id' % (key, order)).fetchall()
    return [(short(n), float(v)) for n, v in rows]


def steps_of(rows, head):
    """Cut at every launch of the step's first kernel; return the segments that show the modal kernel sequence."""
    segs, cur = [], None
    for name, v in rows:
        if name.startswith(head):
            if cur:
                segs.append(cur)
            cur = []
        if cur is not None:
            cur.append((name, v))
    if cur:
        segs.append(cur)
    if not segs:
        raise SystemExit('no launch of %r in the trace' % head)
    sig = collections.Counter(tuple(n for n, _ in s) for s in segs).most_common(1)[0][0]
    return sig, [s for s in segs if tuple(n for n, _ in s) == sig], len(segs)


def per_kernel(steps):
    agg = collections.OrderedDict()
    for s in steps:
        for name, v in s:
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += v
    return agg


def main(argv):
    head = argv[5] if len(argv) > 5 else 'stem_pool_lp'
    sig_f, st_f, nseg_f = steps_of(dispatches(argv[1]), head)
    sig_w, st_w, nseg_w = steps_of(dispatches(argv[2]), head)
    if sig_f != sig_w:
        raise SystemExit('the two passes disagree on the step launch list:\n%s\n%s' % (sig_f, sig_w))
    fetch, write = per_kernel(st_f), per_kernel(st_w)
    by, total = {}, 0.0
    for k, (n, kb) in fetch.items():
        wn, wkb = write[k]
        per_launch = 2.0 * kb / n + wkb / wn
        by[k] = {'launches_per_step': n // len(st_f), 'fetch_kb_raw': round(kb / n, 1), 'write_kb_raw': round(wkb / wn, 1),
                 'hbm_bytes_per_launch': int(per_launch * 1024)}
        total += per_launch * 1024 * n / len(st_f)
    out = {'_meta': {'commit': argv[4] if len(argv) > 4 else '', 'csrc_tree': _csrc_tree(), 'steps_profiled': [len(st_f), len(st_w)],
                     'segments_seen': [nseg_f, nseg_w],
                     'selection': 'dispatch order: segments between launches of %s* with the modal kernel sequence' % head,
                     'command': argv[6] if len(argv) > 6 else 'bench.py --workload backbone_bf16 --steps 20 --min-seconds 0',
                     'correction': 'hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (gfx950: FETCH_SIZE counts 128-byte requests at 64 B)'},
           'hbm_bytes_per_step': int(total), 'launches_per_step': len(sig_f), 'launch_list': list(sig_f), 'by_kernel': by}
    json.dump(out, open(argv[3], 'w'), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == '__main__':
    main(sys.argv)
