"""Dispatches of a window of a rocprofv3 rocpd result with their queue: start offset, duration, queue, kernel.
    python tools/exp/timeline_queues.py x_results.db <first-kernel-substring> [step-index]"""
import re
import sqlite3
import sys


def main(path, first, which=-3):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
    dcols = [r[1] for r in c.execute('pragma table_info(%s)' % disp)]
    name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
    qcol = 'queue_id' if 'queue_id' in dcols else ('stream_id' if 'stream_id' in dcols else 'tid')
    rows = c.execute('select s.%s, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start'
                     % (name_col, qcol, disp, sym)).fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    lo, hi = starts[which], starts[which + 1]
    t0 = rows[lo][1]
    qs = {}
    for n, s, e, q in rows[lo:hi]:
        n = re.sub(r'\s+', ' ', n)
        n = re.sub(r'\(anonymous namespace\)::|samx::', '', n)
        qi = qs.setdefault(q, len(qs))
        print('%8.1f %7.1f  q%d %s%s' % ((s - t0) / 1e3, (e - s) / 1e3, qi, '        ' * qi, n[:60]))
    print('span %.1f us, %d dispatches' % ((rows[hi][1] - t0) / 1e3, hi - lo))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else -3)
