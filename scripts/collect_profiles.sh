#!/bin/bash
# Copy the summaries of one evidence session (scripts/gpu_r05.sh <tag>, merged back into gpurun_out/<tag>/) into profiles/ under a
# round prefix, refresh profiles/pmc_latest.json and regenerate the parity summary.
#   bash scripts/collect_profiles.sh r05b r05
R=gpurun_out/$1; P=profiles/$2
for f in bench.json kernel_stats.csv kernel_stats_C4.csv kernel_stats_C5.csv kernel_stats_S512.csv pmc_hbm_traffic.json pmc_hbm_traffic_S512.json \
         pmc_k6.txt pmc_other.txt configs.jsonl step_sequence.txt step_sequence_shards.txt parity_errors.jsonl k6_numerics.jsonl same_terms.txt \
         two_ranks_one_gpu.log two_ranks_one_gpu_gather.log eight_ranks_one_gpu_60.log eight_ranks_one_gpu_64.log c4_two_ranks_one_gpu.log \
         c4_two_ranks_one_gpu_gather.log stages_S512.log; do
  [ -f $R/$f ] && cp $R/$f ${P}_$f
done
[ -f $R/variants.log ] && cp $R/variants.log ${P}_stages.log
[ -f $R/tests_summary.log ] && cp $R/tests_summary.log ${P}_pytest.log
[ -f $R/pmc_hbm_traffic.json ] && cp $R/pmc_hbm_traffic.json profiles/pmc_latest.json
python scripts/parity_summary.py $R/parity_errors.jsonl $R/k6_numerics.jsonl $R/same_terms.txt > ${P}_parity_summary.md
sed -i 's/[ \t]*$//' ${P}_same_terms.txt 2>/dev/null
ls ${P}_* | wc -l
