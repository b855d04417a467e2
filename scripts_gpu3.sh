python scripts/k6_modes.py 2>&1 | tail -1
timeout 600 python -m pytest tests -m gpu -q -k "backward or fused or high_res or shapenet or determinism or known or headline" 2>&1 | tail -2
