#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x) rocpd SQLite database into a per-kernel stats table
(calls, total / average / min / max duration in microseconds), the equivalent of `--stats` CSV output.

    python scripts/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]
"""
import csv
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in c.execute('pragma table_info(%s)' % disp)]
    scol = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
    name_col = 'display_name' if 'display_name' in scol else 'kernel_name'
    q = ('select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) '
         'from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc' % (name_col, disp, sym, name_col))
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows) or 1
    out = [('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct')]
    for name, n, tot, mn, mx in rows:
        out.append((name[:110], n, '%.1f' % (tot / 1e3), '%.2f' % (tot / n / 1e3), '%.2f' % (mn / 1e3),
                    '%.2f' % (mx / 1e3), '%.1f' % (100.0 * tot / total)))
    w = csv.writer(open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout)
    w.writerows(out)


if __name__ == '__main__':
    main()
