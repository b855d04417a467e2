#!/bin/bash
# round 6: A/B of K6's band kernels.  (1) whole steps on every configuration: the product (k_bpm_row) against k_bpm_fast forced
# (NR_K6_LEGACY=1), default arithmetic and NR_EXACT_GRADIENT=1, two passes of each; (2) the K6 stage call and the fused backward
# through the C ABI over a sweep of shapes (scripts/k6_variants.py), flags 0 / 128 / 2 / 130, plus any variant libraries in the tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-k6ab}; mkdir -p $OUT
for pass in 1 2; do
for leg in row legacy row_exact legacy_exact; do
  unset NR_K6_LEGACY NR_EXACT_GRADIENT
  case $leg in legacy*) export NR_K6_LEGACY=1;; esac
  case $leg in *exact) export NR_EXACT_GRADIENT=1;; esac
  ONLY=${ONLY:-H,SH,C4,C5,X1,X2,X3,X4} timeout 600 python scripts/bench_configs.py 2> $OUT/err_$leg.log | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except ValueError: continue
    if 'config' in d: print('$leg'.ljust(13), d['config'][:60].ljust(60), d.get('ms_fwd_bwd'))
" | tee -a $OUT/ab.txt
done
done
unset NR_K6_LEGACY NR_EXACT_GRADIENT
tail -3 $OUT/err_row.log
VARIANTS="$VARIANTS" K6V_FLAGS="0 128 2 130" SHAPES="${SHAPES:-8x256 16x256 32x256 64x256 128x256 64x320 64x384 64x512 64x640 64x768 32x1024 4x1024 1x2048 256x128 1024x32}" \
  timeout 900 python scripts/k6_variants.py > $OUT/shapes.jsonl 2> $OUT/shapes.err
cat $OUT/shapes.jsonl; tail -3 $OUT/shapes.err
for mode in 10 01; do
  K6V_MODE=$mode K6V_FLAGS="0 128" SHAPES="64x256 64x512" timeout 600 python scripts/k6_variants.py > $OUT/shapes_mode$mode.jsonl 2>> $OUT/shapes.err
  cat $OUT/shapes_mode$mode.jsonl
done
K6V_MESH=ico4 K6V_FLAGS="0 128" SHAPES="64x256 16x256" timeout 600 python scripts/k6_variants.py > $OUT/shapes_ico4.jsonl 2>> $OUT/shapes.err
cat $OUT/shapes_ico4.jsonl
