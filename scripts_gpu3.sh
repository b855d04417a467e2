python scripts/k6_modes.py 2>&1 | tail -1
TAG=now ITERS=10 python scripts/stage_times.py 2>&1 | tail -1 | cut -c75-115
