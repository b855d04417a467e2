timeout 600 python -m pytest tests -m gpu -q -k "backward or fused or high_res or shapenet" 2>&1 | tail -3
TAG=spec ITERS=10 python scripts/stage_times.py 2>&1 | tail -1 | cut -c100-330
