"""Idle time between consecutive kernel dispatches of a rocprofv3 rocpd SQLite result (all queues): where the GPU
waited for the host or for a graph launch.  python profiles/gaps_rocpd.py x_results.db [min_gap_us]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, min_gap=2.0):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
    name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
    rows = c.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start'
                     % (name_col, disp, sym)).fetchall()
    rows = rows[len(rows) // 3:]   # steady state: the last two thirds
    by = defaultdict(lambda: [0, 0.0])
    busy = 0
    prev_end = rows[0][2]
    for n, s, e in rows[1:]:
        busy += e - s
        g = (s - prev_end) / 1e3
        if g >= min_gap:
            k = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'\s+', ' ', n))[:60]
            by[k][0] += 1
            by[k][1] += g
        prev_end = max(prev_end, e)
    span = (rows[-1][2] - rows[0][2]) / 1e3
    print('span %.0f us, kernels busy %.0f us (%.1f %%)' % (span, busy / 1e3, 100 * busy / 1e3 / span))
    for k, (cnt, tot) in sorted(by.items(), key=lambda x: -x[1][1]):
        print('%-60s %5d gaps >= %.0f us before it, %9.1f us in all, %7.1f us each' % (k, cnt, min_gap, tot, tot / cnt))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2.0)
