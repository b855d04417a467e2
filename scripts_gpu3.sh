timeout 600 python -m pytest tests -m gpu -q -k "backward or fused or high_res or shapenet or determinism" 2>&1 | tail -3
TAG=now ITERS=10 python scripts/stage_times.py 2>&1 | tail -1 | cut -c75-115
timeout 600 python scripts/bench_configs.py 2>&1 | tail -3 | cut -c1-220
