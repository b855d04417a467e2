#!/bin/bash
# The multi-GPU curves of BASELINE.md in one command: bench.py --gpus {1,2,4,8} x {teapot, c4} x {no collective,
# --gather}, each launched exactly as the driver launches it (torch.distributed.run, one rank per GPU, RCCL), the JSON lines
# collected into one table with the strong-scaling value, the weak-scaling object and the all-gather column.
#   bash scripts/scale_sweep.sh [out_dir]            on an 8-GPU node
#   GPUS="1 2" ONE_GPU=1 bash scripts/scale_sweep.sh  all ranks on device 0 over gloo (a one-GPU test box: the control path only)
OUT=${1:-gpurun_out/scale}
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
GPUS=${GPUS:-"1 2 4 8"}
WORKLOADS=${WORKLOADS:-"teapot c4"}
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-5}
EXTRA=${EXTRA:-"--cpu-sample-views 0 --light"}
if [ -n "$ONE_GPU" ]; then export NR_DIST_DEVICE=0 NR_DIST_BACKEND=gloo; fi
: > $OUT/lines.jsonl
port=29500
for w in $WORKLOADS; do
  for n in $GPUS; do
    for g in "" "--gather"; do
      [ "$n" = "1" ] && [ -n "$g" ] && continue   # (one rank: nothing to gather)
      port=$((port + 1))
      tag="${w}_n${n}${g:+_gather}"
      if [ "$n" = "1" ]; then
        cmd="python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --workload $w $EXTRA"
      else
        cmd="python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps $STEPS --warmup $WARMUP --workload $w $g $EXTRA"
      fi
      echo "== $tag: $cmd" | tee $OUT/$tag.log
      timeout ${TIMEOUT:-900} $cmd >> $OUT/$tag.log 2>&1
      grep '^{' $OUT/$tag.log | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
if l:
    d = json.loads(l); d['sweep_tag'] = '$tag'; print(json.dumps(d))
" >> $OUT/lines.jsonl
    done
  done
done
python - <<PY
import json
rows = [json.loads(l) for l in open('$OUT/lines.jsonl') if l.strip()]
base = {}
print('| workload | GPUs | collective | ms/step | Mpixel/s (whole job) | vs 1 GPU | weak: Mpixel/s at the same per-GPU size | backend |')
print('|---|---|---|---|---|---|---|---|')
for d in rows:
    w = 'c4' if 'configs[3]' in d['metric'] else 'teapot'
    gather = 'all_gather' in (d['config'].get('parallelism', '') + d['config'].get('workload', ''))
    if d['n_gpus'] == 1:
        base[w] = d['value']
    weak = d.get('weak_scaling') or {}
    print('| %s | %d | %s | %.4f | %.0f | %s | %s | %s |' % (
        w, d['n_gpus'], 'all_gather(rgb)' if gather else 'none', d['ms_per_step'], d['value'],
        ('%.2fx' % (d['value'] / base[w])) if w in base else '-', ('%.0f' % weak['value']) if weak else '-', d['timing'].get('backend')))
PY
