timeout 900 python -m pytest tests -m gpu -q -k "fallback or shapenet or high_res" 2>&1 | tail -12
