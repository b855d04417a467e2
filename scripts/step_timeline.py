#!/usr/bin/env python3
"""One steady-state step of a rocprofv3 --kernel-trace --hip-trace database as a merged timeline: host API calls (HIP
runtime layer) and kernel executions, times in us relative to the step's first kernel.  Development helper.
    python scripts/step_timeline.py x_results.db [anchor-kernel] [step-index]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_face_raster'
    which = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pick = lambda p: [x for x in t if x.startswith(p)]
    disp, sym = pick('rocpd_kernel_dispatch')[0], pick('rocpd_info_kernel_symbol')[0]
    scols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
    name_col = 'display_name' if 'display_name' in scols else 'kernel_name'
    kern = list(c.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start' % (name_col, disp, sym)))
    idx = [i for i, r in enumerate(kern) if anchor in r[0]]
    if len(idx) < which + 2:
        which = len(idx) // 2
    t0, t1 = kern[idx[which]][1], kern[idx[which + 1]][1]
    ev = [(s, 'K', n[:60], e - s) for n, s, e in kern if t0 - 400000 <= s < t1]
    reg = pick('rocpd_region')
    if not reg:
        print('no region table; tables:', t)
    else:
        reg = reg[0]
        rcols = [r[1] for r in c.execute('pragma table_info(%s)' % reg)]
        strt = pick('rocpd_string')[0]
        q = 'select s.string, r.start, r.end, r.tid from %s r join %s s on r.name_id = s.id where r.start >= ? and r.start < ? order by r.start' % (reg, strt)
        try:
            for n, s, e, tid in c.execute(q, (t0 - 400000, t1)):
                ev.append((s, 'H%d' % (tid % 1000), n[:60], e - s))
        except sqlite3.OperationalError as ex:
            print('region query failed', ex, rcols)
    ev.sort()
    for s, kind, n, d in ev:
        print('%9.1f  %-5s %-60s %7.1f' % ((s - t0) / 1e3, kind, n, d / 1e3))


if __name__ == '__main__':
    main()
