#!/bin/bash
# round 5: whole-step A/B of the two band kernels on every configuration: the library's per-launch choice, k_bpm_px forced
# (NR_K6_PX=1), k_bpm_fast forced (NR_K6_LEGACY=1); two passes of each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-k6ab}; mkdir -p $OUT
for pass in 1 2; do
for leg in choice px legacy; do
  unset NR_K6_PX NR_K6_LEGACY
  [ $leg = px ] && export NR_K6_PX=1
  [ $leg = legacy ] && export NR_K6_LEGACY=1
  ONLY=${ONLY:-H,SH,C4,C5,X1,X2,X3,X4} timeout 600 python scripts/bench_configs.py 2> $OUT/err_$leg.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'config' in d: print('$leg'.ljust(7), d['config'][:60].ljust(60), d.get('ms_fwd_bwd'))
" | tee -a $OUT/ab.txt
done
done
tail -3 $OUT/err_choice.log
