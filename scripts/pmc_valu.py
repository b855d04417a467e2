#!/usr/bin/env python3
"""Merge the VALU instruction count of the K6 band kernel (one rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES pass over
scripts/stage_times.py) into a pmc_latest.json as `_valu_issue`, the input of bench.py's secondary roofline.

    python scripts/pmc_valu.py sq_results.db pmc_latest.json [kernel-substring]
"""
import json
import sqlite3
import sys


def main():
    dbs, out = sys.argv[1].split(','), sys.argv[2]  # (several passes: a.db,b.db)
    want = sys.argv[3] if len(sys.argv) > 3 else 'k_bpm_'  # (the band kernel the library picked: the one with the most instructions)
    rec = {}
    for db in dbs:
        c = sqlite3.connect(db)
        t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        pick = lambda p: [x for x in t if x.startswith(p)][0]
        ev, disp, sym, info = pick('rocpd_pmc_event'), pick('rocpd_kernel_dispatch'), pick('rocpd_info_kernel_symbol'), pick('rocpd_info_pmc')
        scols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
        name_col = 'display_name' if 'display_name' in scols else 'kernel_name'
        q = ('select s.%s, i.name, count(distinct d.id), sum(e.value) from %s e join %s d on e.event_id = d.event_id '
             'join %s s on d.kernel_id = s.id join %s i on e.pmc_id = i.id group by s.%s, i.name' %
             (name_col, ev, disp, sym, info, name_col))
        for name, ctr, n, v in c.execute(q):
            if want in name:
                rec.setdefault(name, {})[ctr] = v / max(n, 1)
    if not rec:
        print('no kernel matching', want)
        return
    name, ctrs = sorted(rec.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0))[0]
    try:
        d = json.load(open(out))
    except Exception:
        d = {}
    d['_valu_issue'] = {'kernel': name[:80], 'insts_valu_per_launch': ctrs.get('SQ_INSTS_VALU'),
                        'counters_per_launch': ctrs, 'source': 'rocprofv3 --pmc ' + ' '.join(sorted(ctrs)) + ' -- python scripts/stage_times.py'}
    import os
    rs = os.environ.get('ROW_STATS')  # the output of scripts/row_stats.py for the same shape (work counters of a -DNR_ROW_STATS build)
    if rs and os.path.exists(rs):
        for l in open(rs):
            if l.startswith('{'):
                r = json.loads(l)
                if r.get('B') == 64 and r.get('S') == 256:
                    d['_valu_issue']['row_stats'] = r
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    d.setdefault('_build', {})['csrc_sha1'] = bench.csrc_tree_hash()
    json.dump(d, open(out, 'w'), indent=1)
    print(name[:70], {k: '%.4g' % v for k, v in ctrs.items()})


if __name__ == '__main__':
    main()
