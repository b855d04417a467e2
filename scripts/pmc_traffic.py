#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over scripts/stage_times.py into profiles/pmc_latest.json:
HBM bytes per launch of every kernel and of every C-ABI stage call (the dominant kernel PLUS its helper launches),
corrected as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes: units are KiB, and on gfx950 FETCH_SIZE
counts 64 B per 128 B request of a wide coalesced read, i.e. it reports one half of the bytes of such streams -> the
read side is doubled (the kernels here stage with 16 B/lane loads).

    python scripts/pmc_traffic.py fetch_results.db write_results.db out.json

Attribution.  scripts/stage_times.py (bench.time_stages) issues the stage calls in a fixed order, each 2 + ITERS times
in a row, after ONE set-up forward (the fused forward that produces the upstream gradients).  A kernel that serves
several stages (k_resolve: nr_forward_face_index_map and the fused nr_forward_rasterize; the K6 kernels:
nr_backward_pixel_map and the fused nr_backward_rasterize) is therefore split by dispatch order: its occurrences are
dealt to the stages that launch it, in call order, in equal shares.  If the occurrence count does not fit that protocol
the kernel's overall average is used for each of its stages and the record says so (`attribution: "name"`).
The library's own fills are `nr::k_fill_bytes` launches (the z-buffer in both forward stage calls) and are attributed like every
other kernel; the fused backward's zero fill of grad_textures is part of k_bpm_fast (30 MB of its writes there).
"""
import json
import sqlite3
import sys

# C-ABI stage calls in the order bench.time_stages issues them
CALL_ORDER = ['forward_face_index_map', 'forward_texture_sampling', 'backward_pixel_map', 'backward_textures',
              'backward_depth_map', 'fused_forward_rasterize', 'fused_backward_rasterize']
K6 = ['k_mark_visible', 'k_compact_par', 'k_count_visible', 'k_compact_visible', 'k_band_scan', 'k_band_total',
      'k_line_setup', 'k_bpm_row', 'k_bpm_fast', 'k_bpm_band', 'k_bpm_global', 'k_bpm_finalize']
# kernel-name pattern -> the stage calls that launch it (in CALL_ORDER order); patterns are tried in this order
KERNEL_STAGES = [
    ('k_face_raster', ['forward_face_index_map', 'fused_forward_rasterize']),
    ('k_large_raster', ['forward_face_index_map', 'fused_forward_rasterize']),
    # (late round 4: the kept-workspace forward resolves with k_resolve_quads, the per-call-fill one with k_resolve)
    ('k_resolve_quads', ['fused_forward_rasterize']),
    ('k_resolve', ['forward_face_index_map']),
    # (round 4: the fused forward of bench.time_stages keeps its workspace with epochs like the operator: no fill there; the
    # fused backward's fill rides in k_bpm_fast)
    ('k_fill_bytes', ['forward_face_index_map']),
    ('k_shade', ['forward_texture_sampling']),
] + [(k, ['backward_pixel_map', 'fused_backward_rasterize']) for k in K6] + [
    # (template argument lists are matched as prefixes: the gathers carry a third argument, the per-face light mode)
    ('k_backward_textures_face<true, false', ['backward_textures']),
    ('k_backward_textures_face<false, false', ['backward_textures']),
    ('k_backward_textures_atomic', ['backward_textures']),
    ('k_backward_big<2, false', ['backward_textures']),
    ('k_backward_big<1, false', ['backward_textures']),
    ('k_backward_depth_face', ['backward_depth_map']),
    ('k_backward_big<0, true', ['backward_depth_map']),
    ('k_list_visible', ['backward_depth_map']),
    ('k_backward_textures_face<true, true', ['fused_backward_rasterize']),
    ('k_backward_textures_face<false, true', ['fused_backward_rasterize']),
    ('k_backward_big<2, true', ['fused_backward_rasterize']),
    ('k_backward_big<1, true', ['fused_backward_rasterize']),
]
SETUP_KERNELS = ('k_face_raster', 'k_large_raster', 'k_resolve')  # launched once more by the set-up forward


def dispatches(db, counter):
    """[(dispatch id, kernel name, counter value)] in dispatch order."""
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pick = lambda p: [x for x in t if x.startswith(p)][0]
    ev, disp, sym, info = pick('rocpd_pmc_event'), pick('rocpd_kernel_dispatch'), pick('rocpd_info_kernel_symbol'), pick('rocpd_info_pmc')
    scols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
    name_col = 'display_name' if 'display_name' in scols else 'kernel_name'
    q = ('select d.id, s.%s, sum(e.value) from %s e join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id '
         'join %s i on e.pmc_id = i.id where i.name = ? group by d.id order by d.id' % (name_col, ev, disp, sym, info))
    return list(c.execute(q, (counter,)))


def stages_of(name):
    for pat, st in KERNEL_STAGES:
        if pat in name:
            return pat, st
    return None, None


def attribute(rows, scale):
    """rows: dispatch-ordered (id, name, value).  Returns ({kernel: (launches, mean bytes)}, {stage: {kernel pattern: mean
    bytes per stage call}}, {kernel pattern: 'order' | 'name'})."""
    by_name = {}
    for _, name, v in rows:
        by_name.setdefault(name, []).append(v * scale)
    kernels = {n: (len(v), sum(v) / len(v)) for n, v in by_name.items()}
    stages, how = {}, {}
    for name, vals in by_name.items():
        pat, sts = stages_of(name)
        if pat is None:
            continue
        vals = list(vals)
        if any(k in name for k in SETUP_KERNELS) and len(vals) % len(sts) == 1:
            vals = vals[1:]  # the set-up forward
        if len(vals) % len(sts) == 0 and vals:
            per = len(vals) // len(sts)
            for j, st in enumerate(sts):
                chunk = vals[j * per:(j + 1) * per]
                stages.setdefault(st, {})[pat] = stages.get(st, {}).get(pat, 0.0) + sum(chunk) / len(chunk)
            how[pat] = 'order'
        else:
            for st in sts:
                stages.setdefault(st, {})[pat] = stages.get(st, {}).get(pat, 0.0) + sum(vals) / max(len(vals), 1)
            how[pat] = 'name'
    return kernels, stages, how


def analytic_fills(B, F, S, ts):
    P, N = B * S * S, B * F
    zfill = (P + 1) * 8
    # (up to round 2 the library filled with hipMemsetAsync, whose rocclr kernels cannot be told apart by name; since round 3
    # its fills are nr::k_fill_bytes launches with counters of their own: KERNEL_STAGES lists them, nothing is added here)
    return {}


def main():
    import os
    B, F, S, ts = (int(os.environ.get(k, d)) for k, d in (('B', 64), ('F', 4928), ('S', 256), ('TS', 2)))
    fk, fs, fh = attribute(dispatches(sys.argv[1], 'FETCH_SIZE'), 2.0 * 1024)  # KiB, gfx950 read-side correction x2
    wk, ws, wh = attribute(dispatches(sys.argv[2], 'WRITE_SIZE'), 1024.0)
    out = {}
    fills = analytic_fills(B, F, S, ts)
    for st in CALL_ORDER:
        pats = sorted(set(fs.get(st, {})) | set(ws.get(st, {})))
        if not pats:
            continue
        ks = {p: {'fetch': fs.get(st, {}).get(p, 0.0), 'write': ws.get(st, {}).get(p, 0.0),
                  'attribution': 'order' if fh.get(p) == 'order' and wh.get(p, 'order') == 'order' else 'name'} for p in pats}
        fill = sum(fills.get(st, {}).values())
        out[st] = {'hbm_bytes_per_launch': sum(k['fetch'] + k['write'] for k in ks.values()) + fill,
                   'kernels': ks, 'fills_analytic': fills.get(st, {})}
    kernels = {}
    for name in sorted(set(fk) | set(wk)):
        nf, rd = fk.get(name, (0, 0.0))
        nw, wr = wk.get(name, (0, 0.0))
        kernels[name[:80]] = {'fetch_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'launches': max(nf, nw)}
    out['_kernels'] = kernels
    out['_note'] = ('FETCH_SIZE / WRITE_SIZE in KiB; read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE = 1/2 of wide '
                    'coalesced reads); a stage = every kernel its C-ABI call launches (helpers included) + the library\'s own fills '
                    '(analytic); scene: B %d, F %d, S %d, ts %d' % (B, F, S, ts))
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out['_build'] = {'csrc_sha1': bench.csrc_tree_hash(),
                     'what': 'hash of the library sources these counters were collected on (bench.csrc_tree_hash): bench.py drops the '
                             'records on any other tree'}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k in CALL_ORDER:
        if k in out:
            print('%-28s %.1f MB per launch  (%s)' % (k, out[k]['hbm_bytes_per_launch'] / 1e6, ', '.join(
                '%s %.1f' % (p, (v['fetch'] + v['write']) / 1e6) for p, v in out[k]['kernels'].items())))


if __name__ == '__main__':
    main()
