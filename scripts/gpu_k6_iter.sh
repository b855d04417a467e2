#!/bin/bash
# round 5: k_bpm_px iteration -- K6 tests, error levels + stage times (px vs legacy), kernel trace, SQ counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${TAG:-k6it}; mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ -z "$NOTESTS" ]; then
for f in tests/test_hip_parity.py tests/test_fuzz_gpu.py; do
  echo "=== $f" >> $OUT/pytest.log
  timeout 600 python -m pytest $f -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 >> $OUT/pytest.log
done
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
grep -E "===|passed|failed|error|Error" $OUT/pytest.log | head
fi
VARIANTS="$VARIANTS" SCENES="${SCENES:-H C4}" K6_FLAGS="${K6_FLAGS:-0 128}" ITERS=30 timeout 600 python scripts/k6_numerics.py > $OUT/numerics.jsonl 2> $OUT/numerics.err
python - <<PY
import json
for l in open('$OUT/numerics.jsonl'):
    d = json.loads(l); print(d['scene'], d['variant'], d['flags'], 'us', round(d['stage_us'], 1), 'err', '%.3g' % d['err_floor_metric'])
PY
tail -3 $OUT/numerics.err
VARIANTS="$VARIANTS" SHAPES="${SHAPES:-8x256 16x256 64x256 64x512}" timeout 600 python scripts/k6_variants.py > $OUT/variants.jsonl 2> $OUT/variants.err
cat $OUT/variants.jsonl; tail -3 $OUT/variants.err
if [ -n "$PMC" ]; then
  ITERS=5 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o st -- python scripts/stage_times.py > $OUT/st.log 2>&1
  python scripts/rocpd_stats.py $OUT/st_results.db $OUT/kernel_stats.csv > /dev/null 2>&1
  head -8 $OUT/kernel_stats.csv | cut -c1-60,100-170
  n=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_TRANS SQ_IFETCH" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES"; do
    n=$((n+1))
    ITERS=3 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT -o p$n -- python scripts/stage_times.py > $OUT/p$n.log 2>&1
    python scripts/rocpd_pmc.py $OUT/p${n}_results.db ${KNAME:-k_bpm_row} 2>&1 | cut -c1-30,60-200 | tee -a $OUT/pmc.txt
  done
  rm -f $OUT/*_results.db
fi
