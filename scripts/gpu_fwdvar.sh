#!/bin/bash
for v in "" $VARIANTS; do
  if [ -z "$v" ]; then unset NR_HIP_LIB; else export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so; fi
  echo "== ${v:-base}"
  TAG=${v:-base} ITERS=20 timeout 200 python scripts/stage_times.py 2>/dev/null | tail -1 | cut -c1-120
  TAG=${v:-base} timeout 300 python scripts/profile_config4.py 2>/dev/null | tail -1
  TAG=${v:-base} timeout 300 python scripts/profile_config5.py 2>/dev/null | tail -1
done
