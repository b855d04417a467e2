#!/usr/bin/env python3
"""Where a step's time goes between its kernels: from a rocprofv3 --kernel-trace database, the dispatch sequence of the
steady-state steps (period found from the repetition of the first kernel of the forward) with each kernel's duration and
the idle gap in front of it.

    python scripts/step_gaps.py x_results.db [anchor-kernel-substring]
"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_face_raster'
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [x for x in t if x.startswith('rocpd_kernel_dispatch')][0]
    sym = [x for x in t if x.startswith('rocpd_info_kernel_symbol')][0]
    scols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
    name_col = 'display_name' if 'display_name' in scols else 'kernel_name'
    rows = list(c.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start'
                          % (name_col, disp, sym)))
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    # steady state: the longest run of equal distances (in dispatches) between anchors
    best = (0, 0, 0)
    i = 0
    while i + 1 < len(idx):
        per = idx[i + 1] - idx[i]
        j = i
        while j + 1 < len(idx) and idx[j + 1] - idx[j] == per:
            j += 1
        if j - i > best[0]:
            best = (j - i, i, per)
        i = j if j > i else i + 1
    n, i0, per = best
    if n < 3:
        print('no periodic section found')
        return
    first, last = idx[i0 + 1], idx[i0 + n]  # skip the section's first period
    steps = n - 1
    agg = {}
    for s in range(steps):
        base = first + s * per
        for k in range(per):
            name, st, en = rows[base + k]
            gap = st - rows[base + k - 1][2]
            a = agg.setdefault(k, [name, 0.0, 0.0])
            a[1] += (en - st) / 1e3
            a[2] += gap / 1e3
    span = (rows[last][1] - rows[first][1]) / 1e3 / steps
    print('%d steps of %d dispatches, %.1f us per step' % (steps, per, span))
    tk = tg = 0.0
    for k in range(per):
        name, d, g = agg[k]
        print('%-70s %8.1f us   gap before %6.1f us' % (name[:70], d / steps, g / steps))
        tk += d / steps
        tg += g / steps
    print('kernels %.1f us + gaps %.1f us' % (tk, tg))


if __name__ == '__main__':
    main()
