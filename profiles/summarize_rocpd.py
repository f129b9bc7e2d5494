"""Per-kernel stats (calls, total/avg/min/max ns, share) from a rocprofv3 rocpd SQLite result
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 bench.py ...` writes NAME_results.db).

    python profiles/summarize_rocpd.py gpurun_out/prof/x_results.db > profiles/rNN_x_kernel_stats.txt
"""
import re
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
    name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
    rows = c.execute('select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) '
                     'from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc'
                     % (name_col, disp, sym, name_col)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print('%-90s %8s %14s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns',
                                                  'max_ns', '%'))
    for n, k, t, mn, mx in rows:
        n = re.sub(r'\s+', ' ', n)
        print('%-90s %8d %14d %10.0f %10d %10d %6.2f' % (n[:90], k, t, t / k, mn, mx, 100.0 * t / tot))
    print('TOTAL kernel time ns: %d over %d dispatches' % (tot, sum(r[1] for r in rows)))


if __name__ == '__main__':
    main(sys.argv[1])
