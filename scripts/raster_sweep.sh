#!/bin/bash
# K6 stage call and fused backward over rasters 256 ... 1024 in the three gradient modes: the library's choice, k_bpm_fast forced
# (NR_FLAG_K6_LEGACY = 128) and k_bpm_px forced (NR_FLAG_K6_PX = 65536), 64 teapot views (32 at 1024^2); what the per-launch
# rule of run_backward_pixel_map was read from.   bash scripts/raster_sweep.sh [tag]  ->  gpurun_out/<tag>/raster_sweep.{jsonl,md}
TAG=${1:-sweep}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; : > $OUT/raster_sweep.jsonl
cd scripts
for m in 11 10 01; do
  K6V_MODE=$m ITERS=${ITERS:-15} K6V_FLAGS="0 128 65536" SHAPES="${SHAPES:-64x256 64x320 64x384 64x448 64x512 64x576 64x640 64x768 64x896 32x1024}" \
    timeout 900 python k6_variants.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); d['mode'] = '$m'; print(json.dumps(d))
" >> $OUT/raster_sweep.jsonl
done
python - <<PY > $OUT/raster_sweep.md
import json
rows = [json.loads(l) for l in open('$OUT/raster_sweep.jsonl')]
names = {'11': 'rgb + alpha', '10': 'colour only', '01': 'alpha only'}
print('| gradients | views x raster | K6 stage: choice | k_bpm_fast | k_bpm_px | px / fast | fused backward (all three gradients): choice | k_bpm_fast | k_bpm_px |')
print('|---|---|---|---|---|---|---|---|---|')
for d in rows:
    bwd = ('%.1f | %.1f | %.1f' % (d['product_bwd'], d['product_bwd_f128'], d['product_bwd_f65536'])) if d['mode'] == '11' else '- | - | -'
    print('| %s | %d x %d^2 | %.1f | %.1f | %.1f | %.3f | %s |' % (
        names[d['mode']], d['B'], d['S'], d['product_k6'], d['product_k6_f128'], d['product_k6_f65536'],
        d['product_k6_f65536'] / d['product_k6_f128'], bwd))
PY
cat $OUT/raster_sweep.md
