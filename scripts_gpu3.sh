timeout 900 python -m pytest tests -m gpu -q -k "backward or fused or determinism or headline or known or public" 2>&1 | tail -3
python scripts/k6_modes.py 2>&1 | tail -1
for d in 0 1 4; do NR_K6_DEBUG=$d TAG=k6dbg$d ITERS=10 python scripts/stage_times.py 2>&1 | tail -1 | cut -c75-115; done
