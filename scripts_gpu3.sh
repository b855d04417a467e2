timeout 600 python -m pytest tests -m gpu -q -k "backward or determinism or headline or known" 2>&1 | tail -4
for d in 0 1; do NR_K6_DEBUG=$d TAG=k6dbg$d ITERS=5 python scripts/stage_times.py 2>&1 | tail -1 | cut -c1-140; done
