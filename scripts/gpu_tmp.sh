timeout 900 python -m pytest tests/test_hip_parity.py -q -x -k "backward or big or known or fused or headline or fallback or fast or shapenet or dense or config" 2>&1 | tail -3
ITERS=20 python scripts/stage_times.py 2>&1 | tail -1
