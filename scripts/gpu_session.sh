#!/bin/bash
# One round-3 GPU session (via gpurun): [tests] + stage timings of variant builds + [bench] + [kernel stats] + [PMC].
#   TESTS=1 VARIANTS="old fb5" BENCH=1 KSTATS=1 PMC=1 CONFIGS=1 bash scripts/gpu_session.sh r03a
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ -n "$TESTS" ]; then TEST_TIMEOUT=${TEST_TIMEOUT:-500} bash scripts/gpu_tests.sh $TAG > $OUT/tests_summary.log 2>&1; cat $OUT/tests_summary.log; fi
for v in "" ${VARIANTS}; do
  if [ -z "$v" ]; then unset NR_HIP_LIB; else export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so; fi
  TAG=base$v ITERS=${ITERS:-20} timeout 200 python scripts/stage_times.py 2>&1 | tail -1 >> $OUT/variants.log
  if [ -n "$VARIANTS_AA" ]; then S=512 TAG=S512_base$v ITERS=10 timeout 200 python scripts/stage_times.py 2>&1 | tail -1 >> $OUT/variants.log; fi
done
unset NR_HIP_LIB
for fl in ${STAGE_FLAGS}; do
  NR_STAGE_FLAGS=$fl TAG=flags$fl ITERS=20 timeout 200 python scripts/stage_times.py 2>&1 | tail -1 >> $OUT/variants.log
done
cat $OUT/variants.log
if [ -n "$BENCH" ]; then timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
    print('value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['stages_us'].items()})
    print(json.dumps(d['grad_check'])[:900]); print(json.dumps(d['roofline'])[:1500]); print(d['extra_rows']); print(json.dumps(d['renderer_end_to_end'])); print(d['timing'])
except Exception as e:
    print('bench parse failed', e); print(open('$OUT/bench.err').read()[-3000:])
PY
fi
if [ -n "$CONFIGS" ]; then
  timeout 900 python scripts/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
  ONLY=X1,X2,X3,X4 timeout 600 python scripts/bench_configs.py >> $OUT/configs.jsonl 2>> $OUT/configs.err
  cut -c1-260 $OUT/configs.jsonl; tail -3 $OUT/configs.err
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$KSTATS" ]; then
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o stats -- python bench.py --steps 10 --warmup 2 --cpu-sample-views 0 --light > $OUT/bench_prof.log 2>&1
  python scripts/rocpd_stats.py $OUT/stats_results.db $OUT/kernel_stats.csv > /dev/null 2>&1
  head -16 $OUT/kernel_stats.csv | cut -c1-70,100-170
fi
if [ -n "$STEPSEQ" ]; then
  # the dispatch sequence of a steady-state step with in-step durations and idle gaps: raw C-ABI calls from one thread (the
  # device's own pace) and bench.py's autograd step (host-bound under the tracer: its gaps are the tracer's launch overhead)
  PROBE=A timeout 300 rocprofv3 --kernel-trace -d $OUT -o seqA -- python scripts/thread_gap_probe.py > $OUT/seqA.log 2>&1
  { echo "== raw nr_forward_rasterize + nr_backward_rasterize calls, one thread (scripts/thread_gap_probe.py A)"; python scripts/step_gaps.py $OUT/seqA_results.db k_face_raster; } > $OUT/step_sequence.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace -d $OUT -o seqB -- python bench.py --steps 30 --warmup 3 --light --cpu-sample-views 0 > $OUT/seqB.log 2>&1
  { echo "== bench.py step (autograd operator), traced"; python scripts/step_gaps.py $OUT/seqB_results.db k_face_raster; } >> $OUT/step_sequence.txt 2>&1
  timeout 120 python scripts/thread_gap_probe.py 2>/dev/null | tail -6 >> $OUT/step_sequence.txt
  cat $OUT/step_sequence.txt
fi
if [ -n "$PMC" ]; then
  ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch -- python scripts/stage_times.py > $OUT/fetch.log 2>&1
  ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write -- python scripts/stage_times.py > $OUT/write.log 2>&1
  python scripts/pmc_traffic.py $OUT/fetch_results.db $OUT/write_results.db $OUT/pmc_hbm_traffic.json > $OUT/traffic.log 2>&1
  ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT -o sq -- python scripts/stage_times.py > $OUT/sq.log 2>&1
  python scripts/pmc_valu.py $OUT/sq_results.db $OUT/pmc_hbm_traffic.json >> $OUT/traffic.log 2>&1
  ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $OUT -o sq2 -- python scripts/stage_times.py > $OUT/sq2.log 2>&1
  python scripts/rocpd_pmc.py $OUT/sq_results.db k_bpm > $OUT/pmc_k6.txt 2>&1
  python scripts/rocpd_pmc.py $OUT/sq2_results.db k_bpm >> $OUT/pmc_k6.txt 2>&1
  for k in k_face_raster k_line_setup "k_backward_textures_face<true, true>"; do
    python scripts/rocpd_pmc.py $OUT/sq_results.db "$k" >> $OUT/pmc_other.txt 2>&1
    python scripts/rocpd_pmc.py $OUT/sq2_results.db "$k" >> $OUT/pmc_other.txt 2>&1
  done
  cat $OUT/traffic.log; cat $OUT/pmc_k6.txt | head -40
fi
rm -f $OUT/*_results.db
