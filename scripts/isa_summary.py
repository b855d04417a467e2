#!/usr/bin/env python3
"""Instruction mix / register budget of one kernel from a hipcc --save-temps assembly file (a development aid, CPU only).
    python scripts/isa_summary.py file.s kernel-substring [--dump]
"""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    sub = sys.argv[2]
    for m in re.finditer(r'^(\S*' + re.escape(sub) + r'\S*):.*\n', s, re.M):
        name = m.group(1)
        start = m.end()
        end = s.index('.Lfunc_end', start)
        body = s[start:end]
        lines = [l for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
        c = collections.Counter(l.split()[0] for l in lines)
        kinds = collections.Counter()
        for k, v in c.items():
            kinds['valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else 'lds' if k.startswith('ds_') else 'vmem'] += v
        print(name[:70], 'instructions', len(lines), dict(kinds))
        print('  ', {k: v for k, v in sorted(c.items()) if k.startswith(('s_load', 'global_', 'ds_', 'scratch', 'buffer', 'v_permlane',
                                                                            'v_readfirst', 's_cbranch', 's_barrier', 'v_rcp', 'v_div'))})
        i = s.index('.amdhsa_kernel ' + name)
        blk = s[i:i + 4000]
        print('  ', {k: (re.search(r'\.amdhsa_' + k + r'\s+(\S+)', blk) or [None, None])[1]
                     for k in ('next_free_vgpr', 'next_free_sgpr', 'private_segment_fixed_size', 'group_segment_fixed_size')})
        if '--dump' in sys.argv:
            print(body)


if __name__ == '__main__':
    main()
