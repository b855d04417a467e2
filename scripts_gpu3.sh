for d in 0 1 2 3 4 7; do NR_F2_DEBUG=$d TAG=f2dbg$d ITERS=5 python scripts/stage_times.py 2>&1 | tail -1 | cut -c1-90; done
