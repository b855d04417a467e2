timeout 600 python -m pytest tests -m gpu -q -k "not examples" 2>&1 | tail -4
TAG=fast ITERS=10 python scripts/stage_times.py 2>&1 | tail -1
NR_K6_EXACT=1 TAG=exact ITERS=10 python scripts/stage_times.py 2>&1 | tail -1
