#!/usr/bin/env python3
"""From a rocprofv3 rocpd database: the last N kernel dispatches as a timeline (start, duration, queue, name), to see which
launches of parallel graph branches actually overlapped.   python scripts/rocpd_timeline.py <db> [N]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute('pragma table_info(rocpd_kernel_dispatch)')]
q = 'k.queue_id' if 'queue_id' in cols else '0'
rows = db.execute('select s.display_name, k.start, k.end, %s from rocpd_kernel_dispatch k '
                  'join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start' % q).fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
t0 = rows[0][1]
prev_end = t0
for name, s, e, qid in rows:
    print('%9.1f us  +%7.1f us  q%-3s %s %s' % ((s - t0) / 1e3, (e - s) / 1e3, qid, 'OVERLAP' if s < prev_end else '       ', name[:70]))
    prev_end = max(prev_end, e)
