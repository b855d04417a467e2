"""`bench.py` prints ONE JSON line with the contract's fields (small sizes; the numbers themselves are not asserted)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                          '--batch', '4', '--cpu-sample-views', '1', '--stage-iters', '2'],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['scaling'] == 'strong' and d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert d['config']['views_total'] == 4 and d['config']['views_per_gpu'] == 4 and d['weak_scaling'] is None
    sh = d['shard_rows']  # rank 0's shard of the job at 2 / 4 GPUs, on this GPU (8 GPUs would leave no view: no row)
    assert [r['views'] for r in sh['rows']] == [2, 1] and [p['n_gpus'] for p in sh['predicted_strong_scaling']] == [2, 4]
    assert all(r[k] > 0 for r in sh['rows'] for k in ('ms_autograd', 'ms_autograd_caller_thread', 'ms_function_protocol'))
    assert d['timing']['effective_warmup_steps'] >= d['warmup'] and sh['host_floor']['ms_function_protocol'] > 0
    assert 'workload' in d['config'] and d['value'] > 0 and d['ms_per_step'] > 0
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['achieved'] > 0
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['value'] > 0 and 'sample' in c and c['unit'] == d['unit']
    assert c['all_cores']['value'] > 0 and c['all_cores']['cores'] >= 1 and c['numpy_naive_config1']['value'] > 0
    g = d['grad_check']
    assert g['face_index_mismatch'] == 0
    assert g['grad_faces']['max_abs_err'] <= 1e-4 * g['grad_faces']['max_abs']
    assert g['grad_textures']['max_abs_err'] <= 1e-4 * g['grad_textures']['max_abs']
    assert g['grad_faces']['max_rel_err_floor_1e-3_of_max'] <= 1e-4
    w = r['whole_step']
    assert w['algorithmic_bytes'] == 92 * 4 * 256 * 256 + (108 + 24 * 8) * 4 * d['config']['num_faces']
    # `traffic`: the dominant kernel's own HBM bytes from the committed counter passes -- of the profiled shape (64 views at
    # 256 x 256) only, hence null in this 4-view run; the fields and their scope are on the line all the same
    assert r['traffic'] is None and r['traffic_ratio'] is None and 'traffic_scope' in r and 'traffic' in r['stage_call']
    assert r['kernel'] == 'k_bpm_row'  # (one band kernel in the default mode, whatever the batch size)
    assert d['cold']['ms_per_step'] > 0 and d['exact']['ms_per_step'] > 0 and d['exact']['value'] > 0
    assert len(d['extra_rows']) == 5 and all(x['ms_per_step'] > 0 for x in d['extra_rows'])
    assert any('forward_gpu' in x['row'] for x in d['extra_rows'])
    assert any('NR_FLAG_EXACT_GRADIENT' in x['row'] for x in d['extra_rows'])
    assert d['renderer_end_to_end']['frontend'] == 'fused'
    rows = d['renderer_end_to_end']['reference_protocol']['rows']  # misc/measure_time.py protocol, Renderer defaults (AA on)
    assert [x['batch_size'] for x in rows] == [1, 4, 1] and all(x['anti_aliasing'] and x['raster'] == 512 for x in rows)
    assert all(x[k] > 0 for x in rows for k in ('silhouette_forward_ms', 'silhouette_backward_ms', 'texture_forward_ms',
                                                 'texture_backward_ms'))
    fl = d['renderer_end_to_end']['face_light']  # per-face light colours vs lit, duplicated textures
    assert fl['face_light_ms'] > 0 and fl['lit_textures_ms'] > 0
    assert fl['max_rel_diff']['images'] <= 1e-6 and fl['max_rel_diff']['grad_textures'] <= 1e-5
    st = r['stages']['per_stage']
    assert set(st) == set(d['stages_us']) - {'fused_forward_rasterize', 'fused_backward_rasterize', 'k6_band_kernel_alone',
                                               'k6_band_kernel_alone_in_fused_backward'}
    # the roofline figure is on the dominant kernel alone (events around its launch), the stage call beside it
    assert 0 < r['avg_launch_us'] == d['stages_us']['k6_band_kernel_alone'] < r['stage_call']['avg_us']
    assert r['stage_call']['avg_us'] == d['stages_us']['backward_pixel_map'] and r['in_fused_backward_us'] > 0
    assert all(0 < v['coverage_scaled_bytes'] <= v['algorithmic_bytes'] and v['frac'] < 1.0 for v in st.values())
    assert d['timing']['backend'] is None and d['timing']['rank0_hip_event_ms_per_step'] > 0


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')], cwd=ROOT, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode != 0 and 'needs a GPU' in (out.stderr + out.stdout)
