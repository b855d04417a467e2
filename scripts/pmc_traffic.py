#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_latest.json:
per-kernel HBM bytes per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes:
units are KiB, and on gfx950 FETCH_SIZE counts 64 B per 128 B request of a wide coalesced read, i.e. it reports
one half of the bytes of such streams -> the read side is doubled (the kernels here stage with 16 B/lane loads).

    python scripts/pmc_traffic.py fetch_results.db write_results.db out.json
"""
import json
import sqlite3
import sys

STAGE_OF = {  # kernel-name substring -> bench.py stage key
    'k_bpm_band': 'backward_pixel_map',
    'k_bpm_fast': 'backward_pixel_map',
    'k_compact_small': 'backward_pixel_map',
    'k_face_raster': 'forward_face_index_map',
    'k_resolve': 'forward_face_index_map',
    'k_shade': 'forward_texture_sampling',
    'k_backward_textures_face': 'backward_textures',
    'k_backward_depth_face': 'backward_depth_map',
    'k_mark_visible': 'backward_pixel_map',
    'k_compact_visible': 'backward_pixel_map',
    'k_bpm_finalize': 'backward_pixel_map',
}


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pick = lambda p: [x for x in t if x.startswith(p)][0]
    ev, disp, sym, info = pick('rocpd_pmc_event'), pick('rocpd_kernel_dispatch'), pick('rocpd_info_kernel_symbol'), pick('rocpd_info_pmc')
    scols = [r[1] for r in c.execute('pragma table_info(%s)' % sym)]
    name_col = 'display_name' if 'display_name' in scols else 'kernel_name'
    q = ('select s.%s, count(distinct d.id), sum(e.value) from %s e join %s d on e.event_id = d.event_id '
         'join %s s on d.kernel_id = s.id join %s i on e.pmc_id = i.id where i.name = ? group by s.%s' %
         (name_col, ev, disp, sym, info, name_col))
    return {name: (n, v) for name, n, v in c.execute(q, (counter,))}


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    kernels, stages = {}, {}
    for name in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(name, (0, 0.0))
        nw, vw = write.get(name, (0, 0.0))
        rd = 2.0 * vf * 1024 / max(nf, 1)   # gfx950 correction: x2
        wr = vw * 1024 / max(nw, 1)
        kernels[name[:80]] = {'fetch_bytes_per_launch': rd, 'write_bytes_per_launch': wr, 'launches': max(nf, nw)}
        for sub, stage in STAGE_OF.items():
            if sub in name:
                s = stages.setdefault(stage, {'hbm_bytes_per_launch': 0.0, 'kernels': []})
                s['hbm_bytes_per_launch'] += rd + wr
                s['kernels'].append(sub)
    out = dict(stages)
    out['_kernels'] = kernels
    out['_note'] = 'FETCH_SIZE / WRITE_SIZE in KiB; read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE = 1/2 of wide coalesced reads)'
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k, v in stages.items():
        print('%-28s %.1f MB per launch' % (k, v['hbm_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()
