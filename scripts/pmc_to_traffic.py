#!/usr/bin/env python3
"""Combine the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs because the
TCC block has 4 counter slots) into HBM bytes per launch per conv kernel, with the gfx950
correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 128-byte requests at 64 bytes for
wide (16 B/lane) coalesced reads -> doubled.  Writes profiles/pmc_traffic.json, which bench.py
reads for `roofline.traffic`."""
import json, re, sqlite3, sys

def _csrc_tree():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from usot_amd import build
    return build.csrc_tree()



def per_kernel(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(rocpd_kernel_dispatch)')]
    key = 'event_id' if 'event_id' in cols else 'id'
    rows = db.execute('select s.display_name, p.value from rocpd_pmc_event p '
                      'join rocpd_kernel_dispatch k on k.%s = p.event_id '
                      'join rocpd_info_kernel_symbol s on k.kernel_id = s.id' % key).fetchall()
    agg = {}
    for name, v in rows:
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += v
    return {k: (n, t / n) for k, (n, t) in agg.items()}


def norm(sym):
    m = re.search(r'conv_igemm_f32(_v[23])?<([\d, ]+)>', sym)
    if not m:
        return None
    fam, a = m.group(1) or '', [int(v) for v in m.group(2).split(',')]
    if fam == '':
        return 'conv_igemm_f32<%d,%d>' % (a[0], a[1])
    if fam == '_v3':
        d = a[5] if len(a) > 5 else 1                 # register prefetch depth (tile ids 37-52)
        if len(a) > 7 and a[7] in (4, 5):             # split-fp16 tiles (usot_conv_tile_name: ...,NPW=n,PF=4|5)
            return 'conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=%d,PF=%d>' % (a[0], a[1], a[4], d, a[6], a[7])
        if len(a) > 6 and a[6] == 8:                  # eight producer waves (tile ids 53-60)
            return 'conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=8>' % (a[0], a[1], a[4], d)
        return ('conv_igemm_f32_v3<%d,%d,BK=%d,D=%d>' % (a[0], a[1], a[4], d)) if d > 1 else \
               ('conv_igemm_f32_v3<%d,%d,BK=%d>' % (a[0], a[1], a[4]))
    ksw = a[5] if len(a) > 5 else 1
    if ksw > 1:
        return 'conv_igemm_f32_v2<%d,%d,%d,%d> ksw=%d' % (a[0], a[1], a[4], ksw, ksw)
    return 'conv_igemm_f32_v2<%d,%d,BK=%d>' % (a[0], a[1], a[4])


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
out = {}
for sym, (n, kb) in fetch.items():
    k = norm(sym)
    if k is None:
        continue
    wkb = write.get(sym, (0, 0.0))[1]
    out[k] = {'launches_profiled': n, 'fetch_kb_raw': round(kb, 1), 'write_kb_raw': round(wkb, 1),
              'hbm_bytes_per_launch': int((2.0 * kb + wkb) * 1024)}
out['_meta'] = {'commit': sys.argv[4] if len(sys.argv) > 4 else '', 'csrc_tree': _csrc_tree(), 'command': 'bench.py --steps 30 --min-seconds 0 --no-extras --no-xcorr --no-cpu-baseline',
                'correction': 'hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (gfx950: FETCH_SIZE counts 128-byte requests at 64 B)'}
json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
