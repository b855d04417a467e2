mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample-views 0 > gpurun_out/bench.log 2>&1
tail -12 gpurun_out/pytest.log; tail -2 gpurun_out/smoke.log; python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step']); print(d['stages_us']); print(d['grad_check']); print(d['roofline'])
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/bench.log').read()[-2000:])
PY
