#!/bin/bash
# Copy the summaries of one evidence session (scripts/gpu_r06.sh <tag>, merged back into gpurun_out/<tag>/) into profiles/ under a
# round prefix, refresh profiles/pmc_latest.json and regenerate the parity summary.
#   bash scripts/collect_profiles.sh r06 r06
R=gpurun_out/$1; P=profiles/$2
# (the test session -- TESTS=1 SOAK=... bash scripts/gpu_r06.sh <tag>_tests -- keeps its files in a directory of its own)
for f in parity_errors.jsonl parity_errors_soak.jsonl tests_summary.log; do [ ! -f $R/$f ] && [ -f ${R}_tests/$f ] && cp ${R}_tests/$f $R/$f; done
for f in bench.json kernel_stats.csv kernel_stats_C4.csv kernel_stats_C5.csv kernel_stats_S512.csv pmc_hbm_traffic.json pmc_hbm_traffic_S512.json \
         pmc_k6.txt pmc_other.txt configs.jsonl step_sequence.txt step_sequence_shards.txt parity_errors.jsonl k6_numerics.jsonl same_terms.txt \
         two_ranks_one_gpu.log two_ranks_one_gpu_gather.log eight_ranks_one_gpu_60.log eight_ranks_one_gpu_64.log c4_two_ranks_one_gpu.log \
         c4_two_ranks_one_gpu_gather.log stages_S512.log row_stats.jsonl parity_errors_soak.jsonl rccl_one_rank.log rccl_bench_one_rank.log \
         rccl_bench_one_rank_gather.log; do
  [ -f $R/$f ] && cp $R/$f ${P}_$f
done
[ -f $R/variants.log ] && cp $R/variants.log ${P}_stages.log
[ -f $R/tests_summary.log ] && cp $R/tests_summary.log ${P}_pytest.log
# (the test session's logs land in gpurun_out/ itself: multi-rank and RCCL one-rank runs of tests/test_multi_rank_gpu.py, test_rccl_gpu.py)
for f in two_ranks_one_gpu.log two_ranks_one_gpu_gather.log eight_ranks_one_gpu_60.log eight_ranks_one_gpu_64.log c4_two_ranks_one_gpu.log \
         c4_two_ranks_one_gpu_gather.log rccl_one_rank.log rccl_bench_one_rank.log rccl_bench_one_rank_gather.log; do
  [ ! -f $R/$f ] && [ -f gpurun_out/$f ] && cp gpurun_out/$f ${P}_$f
done
[ -d $R/k6ab ] && python scripts/k6_kernels_md.py $R/k6ab ${P}_k6_kernels.md
if [ -f $R/scale_sweep_table.md ]; then
  { cat <<'MD'
# `scripts/scale_sweep.sh` exercised on the one-GPU test box

`GPUS="1 2" ONE_GPU=1 bash scripts/scale_sweep.sh` -- every rank on device 0, rendezvous and collectives over gloo (RCCL refuses two
ranks on one device).  This is the **control path** of the 8-GPU command (rank -> shard bounds, barrier / max-over-ranks timing, the
`--gather` all-gather of the rendered shards, the weak-scaling leg, one JSON line per run, this table); the two-rank rows share one
GPU and go through host memory for the collective, so they are **not** scaling measurements.  On an 8-GPU node the same script with
its defaults (`GPUS="1 2 4 8"`, RCCL) prints the curve, the weak-scaling column and the all-gather column in one go.  The one-rank
rows are real single-GPU numbers of the `--light` protocol (20 steps behind 5 warm-up steps and the pre-warm).

MD
    cat $R/scale_sweep_table.md; } > ${P}_scale_sweep_one_gpu.md
fi
[ -f $R/pmc_hbm_traffic.json ] && cp $R/pmc_hbm_traffic.json profiles/pmc_latest.json
python scripts/parity_summary.py $R/parity_errors.jsonl $R/k6_numerics.jsonl $R/same_terms.txt > ${P}_parity_summary.md
sed -i 's/[ \t]*$//' ${P}_same_terms.txt 2>/dev/null
ls ${P}_* | wc -l
python scripts/fill_profiles_readme.py > /dev/null
