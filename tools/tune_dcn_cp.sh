#!/bin/bash
# Sweep of the DCN im2col kernel's channels-per-thread constant: rebuilds libvidar_hip.so per variant and times
# `tools/kbench.py dcn`.  Run on a GPU box; restores the default build.  Static side (tools/kernel_resources.py):
# 16 channels = 85 VGPRs, 27 scalar registers parked in VGPR lanes, 5 waves per SIMD; 8 channels = 52 VGPRs, none, 8 waves
# (twice as many per-tap footprints are computed, half as many loads are in flight per wave).
set -u
cd "$(dirname "$0")/.."
for cp in 16 8 4; do
  VIDAR_EXTRA_HIPCC_FLAGS="-DVIDAR_DCN_CP=$cp" python -m vidar_amd.build > /dev/null 2>&1 || { echo "build failed: $cp"; continue; }
  echo "== channels per thread = $cp"
  python tools/kernel_resources.py | grep dcn_im2col_pair
  timeout 200 python tools/kbench.py dcn 2>&1 | grep -i "im2col\|dcn" | cut -c1-140
done
python -m vidar_amd.build > /dev/null 2>&1
echo "default build restored"
