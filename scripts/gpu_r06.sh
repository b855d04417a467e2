#!/bin/bash
# One round-6 GPU session (via gpurun).  Sections by environment switch (round 4's, plus):
#   S512=1      the reference's default raster (anti-aliasing: 512 x 512): kernel stats + FETCH / WRITE counter passes of the stage calls
#   SAMETERMS=1 scripts/same_terms_probe.py (run-to-run differences of the default K6 sums)
#   K6TESTS=1   tests/test_hip_parity.py + tests/test_full_size_gpu.py
#   TESTS=1     every GPU test file (scripts/gpu_tests.sh)
#   VARIANTS="r03 x"  stage timings of libnr_hip_<tag>.so next to the product library, STAGE_FLAGS="2 8" with k6 flags
#   SHARDS="32 16 8"  stage timings + kernel trace + host enqueue time at these batch sizes
#   BENCH=1 / KSTATS=1 / PMC=1 / CONFIGS=1 / STEPSEQ=1   as scripts/gpu_session.sh (BENCH runs behind PMC and reads its counter file)
#   K6AB=1 / SCALE=1   the band kernels side by side; the scaling sweep at 1 and 2 ranks on the one GPU
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ -n "$K6TESTS" ]; then
  rm -f gpurun_out/parity_errors.jsonl
  for f in tests/test_hip_parity.py tests/test_full_size_gpu.py; do
    echo "=== $f" >> $OUT/pytest_k6.log
    timeout 600 python -m pytest $f -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -60 >> $OUT/pytest_k6.log
  done
  cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
  grep -E "===|passed|failed|error|Error|assert" $OUT/pytest_k6.log | head -40
fi
# SOAK=24: the fuzz tests over that many extra seeds (NR_FUZZ_EXTRA_SEEDS; VERDICT r05 item 6: part of every evidence session)
if [ -n "$SOAK" ]; then
  rm -f gpurun_out/parity_errors.jsonl
  NR_FUZZ_EXTRA_SEEDS=$SOAK timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 > $OUT/pytest_soak.log
  cp gpurun_out/parity_errors.jsonl $OUT/parity_errors_soak.jsonl 2>/dev/null
  grep -E "passed|failed|error|assert" $OUT/pytest_soak.log | head
  python scripts/parity_summary.py $OUT/parity_errors_soak.jsonl 2>/dev/null | sed -n '/Cancellation-heavy/,$p' | head -20
fi
if [ -n "$TESTS" ]; then TEST_TIMEOUT=${TEST_TIMEOUT:-500} bash scripts/gpu_tests.sh $TAG > $OUT/tests_summary.log 2>&1; cat $OUT/tests_summary.log; fi
for s in ${SIZES:-256}; do
for v in "" ${VARIANTS}; do
  if [ -z "$v" ]; then unset NR_HIP_LIB; else export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so; fi
  for fl in 0 ${STAGE_FLAGS}; do
    S=$s NR_STAGE_FLAGS=$fl TAG=S${s}_lib${v:-new}_flags$fl ITERS=${ITERS:-20} timeout 200 python scripts/stage_times.py 2>&1 | tail -1 >> $OUT/variants.log
  done
done
done
unset NR_HIP_LIB
[ -f $OUT/variants.log ] && cat $OUT/variants.log
for b in ${SHARDS}; do
  for v in "" ${SHARD_VARIANTS}; do
    if [ -z "$v" ]; then unset NR_HIP_LIB; else export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so; fi
    for fl in 0 ${SHARD_FLAGS}; do
      B=$b NR_STAGE_FLAGS=$fl TAG=B${b}_lib${v:-new}_flags$fl ITERS=30 timeout 200 python scripts/stage_times.py 2>&1 | tail -1 >> $OUT/shards.log
    done
  done
  unset NR_HIP_LIB
  [ -n "$SHARD_PROBE" ] && B=$b timeout 200 python scripts/thread_gap_probe.py 2>/dev/null | tail -6 | sed "s/^/B$b /" >> $OUT/shards.log
done
[ -f $OUT/shards.log ] && cat $OUT/shards.log
if [ -n "$HOSTPROF" ]; then
  for b in $HOSTPROF; do B=$b timeout 300 python scripts/host_profile.py 300 > $OUT/host_profile_B$b.txt 2>&1; head -50 $OUT/host_profile_B$b.txt; done
fi
if [ -n "$CONFIGS" ]; then
  ONLY=${CONFIGS_ONLY} timeout 900 python scripts/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
  cut -c1-300 $OUT/configs.jsonl; tail -3 $OUT/configs.err
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$KSTATS" ]; then
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o stats -- python bench.py --steps 10 --warmup 2 --cpu-sample-views 0 --light > $OUT/bench_prof.log 2>&1
  python scripts/rocpd_stats.py $OUT/stats_results.db $OUT/kernel_stats.csv > /dev/null 2>&1
  head -16 $OUT/kernel_stats.csv | cut -c1-70,100-170
fi
for c in ${KSTATS_CONFIGS}; do
  ONLY=$c timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o stats_$c -- python scripts/bench_configs.py > $OUT/stats_$c.log 2>&1
  python scripts/rocpd_stats.py $OUT/stats_${c}_results.db $OUT/kernel_stats_$c.csv > /dev/null 2>&1
  head -8 $OUT/kernel_stats_$c.csv | cut -c1-70,100-170
done
if [ -n "$NUMERICS" ]; then
  VARIANTS="" SCENES="${NUMERICS}" ITERS=20 timeout 600 python scripts/k6_numerics.py > $OUT/k6_numerics.jsonl 2> $OUT/k6_numerics.err
  cut -c1-330 $OUT/k6_numerics.jsonl
fi
if [ -n "$SHAPES" ]; then
  (cd scripts; ITERS=40 VARIANTS="${SHAPE_VARIANTS}" SHAPES="$SHAPES" timeout 600 python k6_variants.py 2>&1 | grep -v amdgpu.ids) > $OUT/k6_shapes.jsonl
  cat $OUT/k6_shapes.jsonl
fi
for b in ${TRACE_SHARDS}; do
  B=$b PROBE=A timeout 300 rocprofv3 --kernel-trace -d $OUT -o seqB$b -- python scripts/thread_gap_probe.py > $OUT/seqB$b.log 2>&1
  { echo "== B=$b raw nr_forward_rasterize + nr_backward_rasterize calls, one thread"; python scripts/step_gaps.py $OUT/seqB${b}_results.db k_face_raster; } >> $OUT/step_sequence_shards.txt 2>&1
done
[ -f $OUT/step_sequence_shards.txt ] && cat $OUT/step_sequence_shards.txt
if [ -n "$STEPSEQ" ]; then
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
  ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $OUT -o sq2 -- python scripts/stage_times.py > $OUT/sq2.log 2>&1
  # (work counters of the band kernel from a -DNR_ROW_STATS variant build, when the tree holds one: scripts/row_stats.py)
  [ -f neural_renderer_amd/libnr_hip_stats.so ] && SHAPES="64x256 64x512 8x256" timeout 300 python scripts/row_stats.py 2>/dev/null | grep "^{" > $OUT/row_stats.jsonl
  ROW_STATS=$OUT/row_stats.jsonl python scripts/pmc_valu.py $OUT/sq_results.db,$OUT/sq2_results.db $OUT/pmc_hbm_traffic.json >> $OUT/traffic.log 2>&1
  ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $OUT -o sq3 -- python scripts/stage_times.py > $OUT/sq3.log 2>&1
  for db in sq sq2 sq3; do python scripts/rocpd_pmc.py $OUT/${db}_results.db k_bpm >> $OUT/pmc_k6.txt 2>&1; done
  for k in k_face_raster k_line_setup "k_backward_textures_face<true, true>"; do
    for db in sq sq2 sq3; do python scripts/rocpd_pmc.py $OUT/${db}_results.db "$k" >> $OUT/pmc_other.txt 2>&1; done
  done
  cat $OUT/traffic.log; cat $OUT/pmc_k6.txt | head -60
fi
if [ -n "$S512" ]; then
  S=512 ITERS=5 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o st512 -- python scripts/stage_times.py > $OUT/stages_S512.log 2>&1
  python scripts/rocpd_stats.py $OUT/st512_results.db $OUT/kernel_stats_S512.csv > /dev/null 2>&1
  head -14 $OUT/kernel_stats_S512.csv | cut -c1-70,100-170; tail -1 $OUT/stages_S512.log | cut -c1-600
  S=512 ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch512 -- python scripts/stage_times.py > $OUT/fetch512.log 2>&1
  S=512 ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write512 -- python scripts/stage_times.py > $OUT/write512.log 2>&1
  python scripts/pmc_traffic.py $OUT/fetch512_results.db $OUT/write512_results.db $OUT/pmc_hbm_traffic_S512.json > $OUT/traffic_S512.log 2>&1
  cat $OUT/traffic_S512.log | head -30
fi
# K6AB=1: k_bpm_row against k_bpm_fast, both modes: whole steps on every configuration + stage calls over a sweep of shapes
# (scripts/gpu_k6_ab.sh -> profiles/r06_k6_kernels.md through scripts/k6_kernels_md.py)
if [ -n "$K6AB" ]; then TAG=$TAG/k6ab bash scripts/gpu_k6_ab.sh > $OUT/k6ab.log 2>&1; tail -5 $OUT/k6ab.log; fi
# SCALE=1: the turnkey scaling sweep at the rank counts one GPU allows (control-path record: scripts/scale_sweep.sh)
if [ -n "$SCALE" ]; then GPUS="1 2" ONE_GPU=1 timeout 1500 bash scripts/scale_sweep.sh $OUT/scale > $OUT/scale_sweep.log 2>&1; grep "^|" $OUT/scale_sweep.log > $OUT/scale_sweep_table.md; cat $OUT/scale_sweep_table.md; fi
# (after the counter passes, raster 512's included: the session's own counter file becomes profiles/pmc_latest.json of the box's tree, stamped with the tree's
# source hash, so that this bench line carries `roofline.traffic` of the very build it timed)
if [ -n "$BENCH" ]; then [ -f $OUT/pmc_hbm_traffic.json ] && cp $OUT/pmc_hbm_traffic.json profiles/pmc_latest.json; [ -f $OUT/pmc_hbm_traffic_S512.json ] && cp $OUT/pmc_hbm_traffic_S512.json profiles/${TAG%%_*}_pmc_hbm_traffic_S512.json; timeout 900 python bench.py ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
    print('value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 4), {k: round(v, 1) for k, v in d['stages_us'].items()})
    for k in ('grad_check', 'roofline', 'extra_rows', 'shard_rows', 'renderer_end_to_end', 'timing'):
        print(k, json.dumps(d.get(k))[:1500])
except Exception as e:
    print('bench parse failed', e); print(open('$OUT/bench.err').read()[-3000:])
PY
fi
if [ -n "$SAMETERMS" ]; then timeout 600 python scripts/same_terms_probe.py > $OUT/same_terms.txt 2>&1; tail -12 $OUT/same_terms.txt; fi
rm -f $OUT/*_results.db
