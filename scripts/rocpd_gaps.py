#!/usr/bin/env python3
"""From a rocprofv3 rocpd database: busy time vs idle gaps between consecutive kernel
dispatches (all queues merged) over the steady-state part of the trace."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute('select s.display_name, k.start, k.end from rocpd_kernel_dispatch k '
                  'join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start').fetchall()
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rows = rows[-tail:]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
pos = [g for g in gaps if g > 0]
print('dispatches %d  span %.1f us  busy %.1f us (%.1f%%)  idle %.1f us' % (len(rows), span / 1e3, busy / 1e3, 100.0 * busy / span, (span - busy) / 1e3))
print('gap: mean %.2f us  median %.2f us  p90 %.2f us  max %.1f us  (overlapping pairs: %d)' % (
    sum(pos) / len(pos) / 1e3, sorted(pos)[len(pos) // 2] / 1e3, sorted(pos)[int(len(pos) * 0.9)] / 1e3, max(pos) / 1e3, len(gaps) - len(pos)))
