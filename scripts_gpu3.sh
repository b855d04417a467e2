timeout 900 python scripts/bench_configs.py 2>&1 | tail -14 | tee gpurun_out/configs.jsonl
