#!/bin/bash
# One GPU session (via gpurun): tests file by file (a fault in one file must not hide the others), smoke, bench, stage
# timings in both K6 modes, the other BASELINE configurations, rocprofv3 kernel stats; optional PMC passes (PMC=1).
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh r02a'
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/pytest.log
for f in tests/test_hip_parity.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py tests/test_gradient_pins_gpu.py \
         tests/test_sharding_gpu.py tests/test_frontend_gpu.py tests/test_face_light_gpu.py tests/test_texture_io.py tests/test_optimizer.py \
         tests/test_examples_gpu.py tests/test_bench_contract.py tests/test_multi_rank_gpu.py tests/test_rccl_gpu.py; do
  echo "=== $f" >> $OUT/pytest.log
  timeout ${TEST_TIMEOUT:-600} python -m pytest $f -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -${TAIL:-200} >> $OUT/pytest.log
done
echo "=== tests/test_abi.py (no GPU marker: the ABI / symbol checks)" >> $OUT/pytest.log
timeout 300 python -m pytest tests/test_abi.py -q --tb=short -p no:cacheprovider 2>&1 | tail -5 >> $OUT/pytest.log
cp gpurun_out/parity_errors.jsonl gpurun_out/two_ranks_one_gpu*.log $OUT/ 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
TAG=default timeout 300 python scripts/stage_times.py > $OUT/stages.log 2>&1
NR_STAGE_FLAGS=2 TAG=exact timeout 300 python scripts/stage_times.py >> $OUT/stages.log 2>&1
NR_STAGE_FLAGS=8 TAG=scan_path timeout 300 python scripts/stage_times.py >> $OUT/stages.log 2>&1
timeout 300 python scripts/k6_modes.py >> $OUT/stages.log 2>&1
timeout 900 python scripts/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
ONLY=X1,X2,X3,X4 timeout 600 python scripts/bench_configs.py >> $OUT/configs.jsonl 2>> $OUT/configs.err  # extreme shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o stats -- python bench.py --steps 10 --warmup 2 --cpu-sample-views 0 --light > $OUT/bench_prof.log 2>&1
python scripts/rocpd_stats.py $OUT/stats_results.db $OUT/kernel_stats.csv > /dev/null 2>&1
if [ -n "$PMC" ]; then
  ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch -- python scripts/stage_times.py > $OUT/fetch.log 2>&1
  ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write -- python scripts/stage_times.py > $OUT/write.log 2>&1
  python scripts/pmc_traffic.py $OUT/fetch_results.db $OUT/write_results.db $OUT/pmc_latest.json > $OUT/traffic.log 2>&1
  ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT -o sq -- python scripts/stage_times.py > $OUT/sq.log 2>&1
  python scripts/pmc_valu.py $OUT/sq_results.db $OUT/pmc_latest.json >> $OUT/traffic.log 2>&1
  ITERS=3 timeout 600 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $OUT -o sq2 -- python scripts/stage_times.py > $OUT/sq2.log 2>&1
  python scripts/rocpd_pmc.py $OUT/sq_results.db k_bpm > $OUT/pmc_k6.txt 2>&1
  python scripts/rocpd_pmc.py $OUT/sq2_results.db k_bpm >> $OUT/pmc_k6.txt 2>&1
  for k in k_face_raster k_line_setup "k_backward_textures_face<true, true>"; do
    python scripts/rocpd_pmc.py $OUT/sq_results.db "$k" >> $OUT/pmc_other.txt 2>&1
    python scripts/rocpd_pmc.py $OUT/sq2_results.db "$k" >> $OUT/pmc_other.txt 2>&1
  done
fi
rm -f $OUT/*_results.db
grep -E "===|passed|failed|error" $OUT/pytest.log | head -40
tail -2 $OUT/smoke.log
cat $OUT/stages.log
head -14 $OUT/kernel_stats.csv | cut -c1-70,100-170
python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
    print('value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 4)); print({k: round(v, 1) for k, v in d['stages_us'].items()})
    print(json.dumps(d['grad_check'])[:900]); print(json.dumps(d['roofline'])[:700]); print(json.dumps(d['cpu_baseline'])[:900]); print(d['extra_rows'], d['renderer_end_to_end'])
except Exception as e:
    print('bench parse failed', e); print(open('$OUT/bench.err').read()[-3000:])
PY
cat $OUT/configs.jsonl | cut -c1-260; tail -5 $OUT/configs.err
