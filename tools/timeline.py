"""Per-kernel averages and one step's timeline from a rocprofv3 results database (kernel trace).
usage: python tools/timeline.py <results.db> [fraction of the run at which to print one step]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d "
                       f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc").fetchall()
    tot = sum(r[3] for r in rows)
    for r in rows[:18]:
        print(f"{r[0][:72]:72s} n={r[1]:6d} avg={r[2] / 1000:9.2f}us {100 * r[3] / tot:5.1f}%")
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s "
                       f"on d.kernel_id=s.id order by d.start").fetchall()
    i0 = int(len(rows) * frac)
    while 'apply' not in rows[i0][0]:
        i0 += 1
    t0 = rows[i0][2]
    j, cnt = i0 + 1, 0
    while cnt < 2 and j < len(rows):
        r = rows[j]
        print(f"{(r[1] - t0) / 1000:8.2f} {(r[2] - t0) / 1000:8.2f} q{r[3]} {r[0][:80]}")
        cnt += 'apply' in r[0]
        j += 1


if __name__ == '__main__':
    main()
