"""Writes the figures of profiles/README.md's "Recomputing the round-6 bench line" section (between its figures:begin / figures:end
marks, from the template below) and the r06 row's headline from profiles/r06_bench.json, so that the text and the committed line
cannot drift apart:   python scripts/fill_profiles_readme.py   (scripts/collect_profiles.sh runs it)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(os.path.join(ROOT, 'profiles', 'r06_bench.json')).read().strip().splitlines()[-1])
r, sc = d['roofline'], d['roofline']['stage_call']
rows = d['shard_rows']['rows']
aa = d['extra_rows'][0]['roofline']
rep = {
    'VALUE_MS4': '%.4f' % d['ms_per_step'], 'VALUE_MS': '%.5f' % d['ms_per_step'], 'VALUE_MP': '{:,.0f}'.format(d['value']).replace(',', ' '),
    'HIP_MS': '%.5f' % d['timing']['rank0_hip_event_ms_per_step'], 'COLD_MS': '%.5f' % d['cold']['ms_per_step'],
    'EXACT_MS': '%.5f' % d['exact']['ms_per_step'], 'AVG_US': '%.2f' % r['avg_launch_us'], 'FUSED_US': '%.1f' % r['in_fused_backward_us'],
    'STAGE_ACH': '%.0f' % sc['achieved'], 'STAGE_FRAC': '%.3f' % sc['frac'], 'STAGE_US': '%.1f' % sc['avg_us'],
    'ACH': '{:,.0f}'.format(r['achieved']).replace(',', ' '), 'FRAC': '%.3f' % r['frac'],
    'STR_MB': '%.1f' % (sc['traffic'] / 1e6), 'STR_RATIO': '%.2f' % sc['traffic_ratio'],
    'TR_MB': '%.1f' % (r['traffic'] / 1e6), 'TR_RATIO': '%.2f' % r['traffic_ratio'],
    'AA_MB': '%.1f' % (aa['traffic'] / 1e6), 'AA_RATIO': '%.2f' % aa['traffic_ratio'],
    'STAMP': r['traffic_source']['stamp']['file_csrc_sha1'][:8],
    'LANE_EXEC': '%.3f' % r['lane_efficiency']['executing_lanes_per_issued_lane'],
    'SH_A': ' / '.join('%.4f' % x['ms_autograd'] for x in rows), 'SH_F': ' / '.join('%.4f' % x['ms_function_protocol'] for x in rows),
    'SH_U': ' / '.join('%.1f' % x['roofline']['fused_calls_us'] for x in rows),
    'PRED': ' / '.join('%.1f' % (p['value_autograd'] / 1e3) for p in d['shard_rows']['predicted_strong_scaling']),
}
TEMPLATE = '''* `value` = 64·256² / `ms_per_step` (`r06_bench.json`: VALUE_MS ms → VALUE_MP Mpixel/s; HIP events on the launch stream:
  `timing.rank0_hip_event_ms_per_step` HIP_MS; 100 timed steps behind 250 ms of untimed pre-warm and 30 warm-up steps).  `cold`
  COLD_MS ms; `exact` EXACT_MS ms.  (The round's other sessions, other boxes and earlier trees: 0.2959–0.2999 ms.)  The step is 9
  launches; inside a steady-state step (`r06_step_sequence.txt`, the traced `bench.py`): `k_face_raster` + `k_large_raster` +
  `k_resolve_quads` ≈ 77 µs, `k_compact_par` 11 + `k_line_setup` 28 + `k_bpm_row` 129 (with the grad_textures fill inside) +
  `k_bpm_fast`'s overflow-only launch 4.7 + fused gather 45 + `k_backward_big` 4.6 ≈ 223 µs, + ~5 µs in front of the first launch.
  `r06_kernel_stats.csv` has the same averages.
* `roofline.avg_launch_us` AVG_US µs: `k_bpm_row` ALONE inside the staged call `nr_backward_pixel_map`, HIP events recorded by the
  measurement build of the library right in front of and behind that launch; `in_fused_backward_us` FUSED_US.  `achieved` =
  173 703 168 B ÷ AVG_US µs = ACH GB/s; `frac` = ÷ 8000 = **FRAC** (round 5: 0.130).  `stage_call` STAGE_US µs → STAGE_ACH GB/s, STAGE_FRAC.
* `roofline.traffic`: `k_bpm_row` **TR_MB MB, TR_RATIO×** of its algorithmic bytes; the stage call **STR_MB MB, STR_RATIO×** (`k_line_setup`
  ~88, `k_bpm_finalize` ~16, `k_compact_par` ~13: `r06_pmc_hbm_traffic.json`).  At raster 512² (`_S512`; the anti-aliasing row's
  `roofline.traffic`): `k_bpm_row` AA_MB MB against 626.7 (AA_RATIO×), stage call ~910 MB (1.45×).  The measurement session collects the
  counters (both rasters) first and hands the files — stamped with the hash of the sources — to its own bench run
  (`traffic_source.stale: false`, both stamps `STAMP…`); a run on a tree whose sources differ from the counter files' prints
  `traffic: null`.
* `roofline.lane_efficiency`: LANE_EXEC executing lanes per issued lane (`SQ_THREAD_CYCLES_VALU` ÷ 64 ÷ `SQ_INSTS_VALU`, `r06_pmc_k6.txt`),
  0.893 of the visits' lanes on a pixel of their sweep (`r06_row_stats.jsonl`).
* `shard_rows`: SH_A ms (`ms_autograd`; `ms_function_protocol` SH_F), each with its
  `roofline` (`fused_calls_us` SH_U); `predicted_strong_scaling` PRED Gpixel/s.
'''
import re
text = TEMPLATE
for k in sorted(rep, key=len, reverse=True):
    text = text.replace(k, rep[k])
p = os.path.join(ROOT, 'profiles', 'README.md')
s = open(p).read()
a = s.index('<!-- figures:begin (scripts/fill_profiles_readme.py) -->') + len('<!-- figures:begin (scripts/fill_profiles_readme.py) -->\n')
b = s.index('<!-- figures:end -->')
s = s[:a] + text + s[b:]
s = re.sub(r'`r06_bench.json` \(\*\*[^*]*\*\*; round 5: 12 296\)',
           '`r06_bench.json` (**%s ms = %s Mpixel/s**; round 5: 12 296)' % (rep['VALUE_MS4'], rep['VALUE_MP']), s)
open(p, 'w').write(s)
print({k: rep[k] for k in ('VALUE_MS', 'VALUE_MP', 'AVG_US', 'FRAC', 'TR_RATIO', 'STAMP')})
