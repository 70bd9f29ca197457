#!/bin/bash
# bucket-index bits per window (tuning knob msm_c) against MSM time, G1 and G2 (measurement tool)
for c in 15 16 17 18 19 20; do
  echo "== msm_c $c"
  ZP_MSM_C=$c ZP_MSM_G2="${G2SIZES:-22}" python tools/msm_bench.py ${G1SIZES:-22 24}
done
