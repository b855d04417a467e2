timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample-views 0 2> gpurun_out/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['launch_mode'], d['eager_ms_per_step']); print(d['stages_us']); print(d['grad_check'])"
tail -3 gpurun_out/bench.err
