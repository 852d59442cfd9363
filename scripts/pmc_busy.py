#!/usr/bin/env python3
"""Sustained clock and matrix-pipe utilisation per kernel from ONE `rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES
SQ_VALU_MFMA_BUSY_CYCLES` pass (rocpd database).  rocprofv3 reports one value per shader engine (32 on gfx950) per dispatch:

    clock [GHz]     = mean-over-SEs SQ_BUSY_CYCLES / dispatch duration [ns]
    mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs per SE x SQ_BUSY_CYCLES),   SIMDs per SE = 256 CUs x 4 / 32 = 32

(SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, summed over the SE's SIMDs: MI355X_MICROARCH.md, 'per-instruction cycle
constants'.)  Prints the per-kernel table (what profiles/roundN_*_mfma_busy.txt holds) and, with an output path, merges
{kernel: {clock_ghz, mfma_busy_frac, avg_us, calls}} into a JSON file (profiles/pmc_busy.json) that bench.py reads for
`peak_sustained`.  Kernel keys: conv_igemm_f32 tiles as hip.tile_name() prints them, everything else the demangled symbol
up to its argument list.
    python scripts/pmc_busy.py RESULTS.db [out.json [commit [min_us]]]"""
import json, os, re, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_lp_traffic import short


def f32_tile_name(sym):
    """The key bench.py uses for an fp32 conv tile (usot_conv_tile_name), from the kernel symbol; None for other kernels."""
    m = re.search(r'conv_igemm_f32(_v[23])?<([\d, ]+)>', sym)
    if not m:
        return None
    fam, a = m.group(1) or '', [int(v) for v in m.group(2).split(',')]
    if fam == '':
        return 'conv_igemm_f32<%d,%d>' % (a[0], a[1])
    if fam == '_v3':
        d = a[5] if len(a) > 5 else 1
        if len(a) > 7 and a[7] in (4, 5):             # split-fp16 tiles (usot_conv_tile_name: ...,NPW=n,PF=4|5)
            return 'conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=%d,PF=%d>' % (a[0], a[1], a[4], d, a[6], a[7])
        if len(a) > 6 and a[6] == 8:
            return 'conv_igemm_f32_v3<%d,%d,BK=%d,D=%d,NPW=8>' % (a[0], a[1], a[4], d)
        return ('conv_igemm_f32_v3<%d,%d,BK=%d,D=%d>' % (a[0], a[1], a[4], d)) if d > 1 else ('conv_igemm_f32_v3<%d,%d,BK=%d>' % (a[0], a[1], a[4]))
    ksw = a[5] if len(a) > 5 else 1
    if ksw > 1:
        return 'conv_igemm_f32_v2<%d,%d,%d,%d> ksw=%d' % (a[0], a[1], a[4], ksw, ksw)
    return 'conv_igemm_f32_v2<%d,%d,BK=%d>' % (a[0], a[1], a[4])


def main(argv):
    db = sqlite3.connect(argv[1])
    cols = [r[1] for r in db.execute('pragma table_info(rocpd_kernel_dispatch)')]
    key = 'event_id' if 'event_id' in cols else 'id'
    rows = db.execute('select k.id, s.display_name, k.end - k.start, i.name, p.value from rocpd_pmc_event p '
                      'join rocpd_info_pmc i on p.pmc_id = i.id '
                      'join rocpd_kernel_dispatch k on k.%s = p.event_id '
                      'join rocpd_info_kernel_symbol s on k.kernel_id = s.id' % key).fetchall()
    disp = {}
    for did, sym, dur, ctr, v in rows:
        d = disp.setdefault(did, {'sym': sym, 'dur': dur, 'c': {}})
        c = d['c'].setdefault(ctr, [0, 0.0])
        c[0] += 1
        c[1] += v
    agg = {}
    for d in disp.values():
        if 'SQ_BUSY_CYCLES' not in d['c'] or 'SQ_VALU_MFMA_BUSY_CYCLES' not in d['c']:
            continue
        name = f32_tile_name(d['sym']) or short(d['sym'])
        n_se, busy = d['c']['SQ_BUSY_CYCLES']
        _, mfma = d['c']['SQ_VALU_MFMA_BUSY_CYCLES']
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, n_se])
        a[0] += 1
        a[1] += d['dur']
        a[2] += busy / n_se                      # cycles the SE's sequencers were busy (mean over SEs) = elapsed shader cycles
        a[3] += mfma / n_se                      # MFMA-busy cycles per SE, summed over its SIMDs
    simds_per_se = 256 * 4 // 32
    min_us = float(argv[4]) if len(argv) > 4 else 0.0
    out = {}
    print('%-84s %7s %10s %10s %10s' % ('kernel', 'calls', 'avg_us', 'clock_GHz', 'mfma_busy'))
    for name, (n, dur, busy, mfma, n_se) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if dur / n / 1e3 < min_us:
            continue
        e = {'calls': n, 'avg_us': round(dur / n / 1e3, 2), 'clock_ghz': round(busy / dur, 3),
             'mfma_busy_frac': round(mfma / (simds_per_se * busy), 4), 'instances_per_dispatch': n_se}
        out[name] = e
        print('%-84s %7d %10.2f %10.3f %10.4f' % (name, n, e['avg_us'], e['clock_ghz'], e['mfma_busy_frac']))
    if len(argv) > 2 and argv[2]:
        path = argv[2]
        cur = json.load(open(path)) if os.path.exists(path) else {'_meta': {}, 'kernels': {}}
        cur['kernels'].update(out)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from usot_amd import build as _b
        cur['_meta'].update({'commit': argv[3] if len(argv) > 3 else '', 'csrc_tree': _b.csrc_tree(),
                             'what': 'rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES; clock = SQ_BUSY_CYCLES (mean over '
                                     'the 32 SEs) / duration; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (32 SIMDs per SE x SQ_BUSY_CYCLES)'})
        json.dump(cur, open(path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main(sys.argv)
