#!/bin/bash
for leg in 0 1; do
NR_K6_LEGACY=$leg python bench.py --cpu-sample-views 0 --no-shard-rows 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('legacy=$leg', 'ms', round(d['ms_per_step'], 4), 'cold', round(d['cold']['ms_per_step'], 4), [(x['row'][:20], round(x['ms_per_step'], 4)) for x in d['extra_rows']])
"
done
