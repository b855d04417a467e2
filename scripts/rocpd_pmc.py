#!/usr/bin/env python3
"""Per-kernel sums of the PMC counters stored in a rocprofv3 rocpd SQLite database.
    python scripts/rocpd_pmc.py x_results.db [kernel-substring]
"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pick = lambda p: [x for x in t if x.startswith(p)][0]
    ev, disp, sym, info = pick('rocpd_pmc_event'), pick('rocpd_kernel_dispatch'), pick('rocpd_info_kernel_symbol'), pick('rocpd_info_pmc')
    ecols = [r[1] for r in c.execute('pragma table_info(%s)' % ev)]
    icols = [r[1] for r in c.execute('pragma table_info(%s)' % info)]
    scols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
    dcols = [r[1] for r in c.execute('pragma table_info(%s)' % disp)]
    name_col = 'display_name' if 'display_name' in scols else 'kernel_name'
    q = ('select s.%s, i.name, count(*), sum(e.value) from %s e join %s d on e.event_id = d.event_id '
         'join %s s on d.kernel_id = s.id join %s i on e.pmc_id = i.id group by s.%s, i.name' %
         (name_col, ev, disp, sym, info, name_col))
    try:
        rows = list(c.execute(q))
    except sqlite3.OperationalError as ex:
        print('query failed:', ex)
        print('pmc_event cols', ecols, '\ninfo_pmc cols', icols, '\ndispatch cols', dcols)
        return
    for name, ctr, n, v in sorted(rows):
        if flt in name:
            print('%-60s %-28s n=%-5d sum=%.4g per_dispatch=%.4g' % (name[:60], ctr, n, v, v / max(n, 1)))


if __name__ == '__main__':
    main()
