"""Writes the measured tables of profiles/r06_k6_kernels.md (between its `<!-- tables:begin -->` / `<!-- tables:end -->` marks) from
the files of one `scripts/gpu_k6_ab.sh` session:   python scripts/k6_kernels_md.py gpurun_out/k6ab profiles/r06_k6_kernels.md"""
import collections
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
out = []

rows = collections.OrderedDict()
for ln in open(os.path.join(src, 'ab.txt')):
    leg, cfg, val = ln[:13].strip(), ln[14:74].strip(), ln[74:].strip()
    rows.setdefault(cfg, collections.OrderedDict()).setdefault(leg, []).append(val)
out.append('## Whole steps (`Rasterize` forward + backward through the autograd operator, `scripts/bench_configs.py`, ms, two passes)\n')
out.append('| configuration | k_bpm_row | k_bpm_fast (`NR_K6_LEGACY=1`) | exact mode: k_bpm_row | exact mode: k_bpm_fast |')
out.append('|---|---|---|---|---|')
for cfg, d in rows.items():
    cells = [', '.join(d.get(k, [])) for k in ('row', 'legacy', 'row_exact', 'legacy_exact')]
    out.append('| %s | %s |' % (cfg, ' | '.join(cells)))


def load(name):
    p = os.path.join(src, name)
    return [json.loads(l) for l in open(p) if l.startswith('{')] if os.path.exists(p) else []


def table(title, recs, flag_sets):
    if not recs:
        return
    out.append('\n' + title + '\n')
    out.append('| shape | ' + ' | '.join(h for h, _ in flag_sets) + ' |')
    out.append('|---|' + '---|' * len(flag_sets))
    for r in recs:
        cells = []
        for _, suffix in flag_sets:
            k6, bwd = r.get('product_k6' + suffix), r.get('product_bwd' + suffix)
            cells.append('%s / %s' % (k6, bwd) if k6 is not None else '-')
        out.append('| %d x %d^2 | %s |' % (r['B'], r['S'], ' | '.join(cells)))


four = (('k_bpm_row', ''), ('k_bpm_fast', '_f128'), ('exact: k_bpm_row', '_f2'), ('exact: k_bpm_fast', '_f130'))
two = four[:2]
table('## K6 stage call / fused backward (C ABI, HIP events, `scripts/k6_variants.py`, us; teapot views, rgb + alpha (+ depth in the '
      'fused backward))', load('shapes.jsonl'), four)
table('## The same with the colour gradient only (K6 stage call / fused backward, us)', load('shapes_mode10.jsonl'), two)
table('## ... and with the alpha gradient only', load('shapes_mode01.jsonl'), two)
table('## Dense meshes: icospheres of 10 240 faces (fill_back), random rotations', load('shapes_ico4.jsonl'), two)

text = open(dst).read()
a, b = text.index('<!-- tables:begin -->') + len('<!-- tables:begin -->'), text.index('<!-- tables:end -->')
open(dst, 'w').write(text[:a] + '\n' + '\n'.join(out) + '\n' + text[b:])
