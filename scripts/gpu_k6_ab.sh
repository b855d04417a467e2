#!/bin/bash
# round 5: whole-step A/B of the two band kernels (NR_K6_LEGACY) on every configuration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-k6ab}; mkdir -p $OUT
for leg in 0 1 0 1; do
  NR_K6_LEGACY=$leg ONLY=H,SH,C4,C5,X1,X2,X3,X4 timeout 600 python scripts/bench_configs.py 2> $OUT/err_$leg.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'config' in d: print('legacy=$leg', d['config'][:60].ljust(60), d.get('ms_fwd_bwd'))
" | tee -a $OUT/ab.txt
done
tail -3 $OUT/err_0.log
