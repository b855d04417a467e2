#!/bin/bash
# Run on the GPU box (via gpurun): tests, smoke, bench, rocprofv3 kernel stats and HBM-traffic PMC passes.
set -x
TAG=${1:-r01}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/$TAG/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG -o stats -- python bench.py --steps 10 --warmup 2 --cpu-sample-views 0 > gpurun_out/$TAG/bench_prof.log 2>&1
python scripts/rocpd_stats.py gpurun_out/$TAG/stats_results.db gpurun_out/$TAG/kernel_stats.csv
ITERS=3 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/$TAG -o fetch -- python scripts/stage_times.py > gpurun_out/$TAG/fetch.log 2>&1
ITERS=3 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/$TAG -o write -- python scripts/stage_times.py > gpurun_out/$TAG/write.log 2>&1
python scripts/pmc_traffic.py gpurun_out/$TAG/fetch_results.db gpurun_out/$TAG/write_results.db gpurun_out/$TAG/pmc_latest.json
rm -f gpurun_out/$TAG/*_results.db
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -3 gpurun_out/$TAG/pytest.log; tail -1 gpurun_out/$TAG/smoke.log; head -12 gpurun_out/$TAG/kernel_stats.csv | cut -c1-60,105-190; cat gpurun_out/$TAG/bench.json | cut -c1-1500
