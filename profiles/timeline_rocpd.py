"""Kernel timeline of one steady-state step from a rocprofv3 rocpd SQLite result: per dispatch the
duration and the idle gap since the previous dispatch ended (same process, all queues).

    python profiles/timeline_rocpd.py x_results.db <first-kernel-substring> [step-index]
"""
import re
import sqlite3
import sys


def main(path, first, which=-3):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
    name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
    rows = c.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start'
                     % (name_col, disp, sym)).fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    lo = starts[which]
    hi = starts[which + 1]
    prev_end = rows[lo - 1][2]
    tot_k = tot_gap = 0
    for n, s, e in rows[lo:hi]:
        n = re.sub(r'\s+', ' ', n)
        n = re.sub(r'\(anonymous namespace\)::', '', n)
        print('%-70s dur %7.2f us  gap %7.2f us' % (n[:70], (e - s) / 1e3, (s - prev_end) / 1e3))
        tot_k += e - s
        tot_gap += max(0, s - prev_end)
        prev_end = max(prev_end, e)
    print('step: %d dispatches, kernel time %.1f us, gaps %.1f us, span %.1f us'
          % (hi - lo, tot_k / 1e3, tot_gap / 1e3, (rows[hi][1] - rows[lo][1]) / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else -3)
