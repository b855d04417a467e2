mkdir -p gpurun_out
python -c "import torch;print(torch.cuda.get_device_name(0))" > gpurun_out/dev.log 2>&1
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -80 > gpurun_out/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1
tail -5 gpurun_out/pytest.log; tail -3 gpurun_out/smoke.log; tail -c 3000 gpurun_out/bench.log
