#!/bin/bash
# kernel stats of several variant builds, one rocprofv3 pass each
for v in "" $VARIANTS; do
  if [ -z "$v" ]; then unset NR_HIP_LIB; else export NR_HIP_LIB=$PWD/neural_renderer_amd/libnr_hip_$v.so; fi
  bash scripts/gpu_kstats.sh kvar_${v:-base} > /dev/null 2>&1
  echo "== ${v:-base}"; grep -E "k_bpm_fast|k_line_setup|k_compact|k_face_raster|k_backward_textures_face<true, true>" gpurun_out/kvar_${v:-base}/kernel_stats.csv | cut -c1-60,112-150
done
