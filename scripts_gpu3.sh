timeout 600 python -m pytest tests -m gpu -q -k "backward or fused or high_res or shapenet or determinism or known or headline" 2>&1 | tail -2
python scripts/k6_modes.py 2>&1 | tail -1
TAG=now ITERS=10 python scripts/stage_times.py 2>&1 | tail -1 | cut -c75-115
