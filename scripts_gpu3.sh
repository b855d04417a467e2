timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3
TAG=now ITERS=10 python scripts/stage_times.py 2>&1 | tail -1
