#!/usr/bin/env python3
"""HBM traffic of the batch-64 low-precision backbone step (BASELINE configs[2]) from two rocprofv3 --pmc passes of
`bench.py --workload backbone_bf16` (FETCH_SIZE, WRITE_SIZE: separate runs, the TCC block has 4 counter slots), with the gfx950
correction of MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests at 64 bytes for wide coalesced reads -> doubled.
Writes profiles/pmc_traffic_bf16.json: bytes per STEP (one stem_pool_lp launch = one step) and per launch of every kernel
symbol; bench.py reads it for `backbone_bf16_b64.roofline.traffic`.
    python scripts/pmc_lp_traffic.py FETCH.db WRITE.db out.json [commit]"""
import json, re, sqlite3, sys


def short(name):
    name = re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*$', '', name)[:80]


def per_kernel(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(rocpd_kernel_dispatch)')]
    key = 'event_id' if 'event_id' in cols else 'id'
    rows = db.execute('select s.display_name, p.value from rocpd_pmc_event p '
                      'join rocpd_kernel_dispatch k on k.%s = p.event_id '
                      'join rocpd_info_kernel_symbol s on k.kernel_id = s.id' % key).fetchall()
    agg = {}
    for name, v in rows:
        a = agg.setdefault(short(name), [0, 0.0])
        a[0] += 1
        a[1] += v
    return agg


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
is_step_kernel = lambda k: k.startswith('conv_igemm_bf16') or k.startswith('pw_panel_kernel') or k.startswith('pw_pair_kernel') or k.startswith('stem_pool_lp')
steps_f = sum(n for k, (n, _) in fetch.items() if k.startswith('stem_pool_lp'))
steps_w = sum(n for k, (n, _) in write.items() if k.startswith('stem_pool_lp'))
by, total = {}, 0.0
for k, (n, kb) in fetch.items():
    if not is_step_kernel(k):
        continue
    wn, wkb = write.get(k, (0, 0.0))
    per_launch = 2.0 * kb / n + (wkb / wn if wn else 0.0)
    by[k] = {'launches_per_step': round(n / steps_f, 3), 'fetch_kb_raw': round(kb / n, 1), 'write_kb_raw': round(wkb / wn, 1) if wn else 0.0,
             'hbm_bytes_per_launch': int(per_launch * 1024)}
    total += per_launch * 1024 * n / steps_f
out = {'_meta': {'commit': sys.argv[4] if len(sys.argv) > 4 else '', 'steps_profiled': [steps_f, steps_w],
                 'command': 'bench.py --workload backbone_bf16 --steps 20 --min-seconds 0',
                 'correction': 'hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (gfx950: FETCH_SIZE counts 128-byte requests at 64 B)'},
       'hbm_bytes_per_step': int(total), 'by_kernel': by}
json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
