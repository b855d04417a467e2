timeout 600 python -m pytest tests -m gpu -q -k "backward or fused" 2>&1 | tail -3
for kb in 80 53 48 40; do NR_K6_LDS_KB=$kb TAG=lds$kb ITERS=10 python scripts/stage_times.py 2>&1 | tail -1 | cut -c60-150; done
