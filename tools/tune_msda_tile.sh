#!/bin/bash
# Tuning sweep of the msda_bwd accumulate kernel's compile-time constants: rebuilds libvidar_hip.so per variant
# and times `tools/kbench.py msda`.  Run on a GPU box; restores the default build.
#   VIDAR_MSDA_TWAVES waves (= private LDS windows) per workgroup, VIDAR_MSDA_CHUNK records per chunk
set -u
cd "$(dirname "$0")/.."
for v in "7 1024" "4 1024" "7 512" "7 2048" "14 1024"; do
  set -- $v
  VIDAR_EXTRA_HIPCC_FLAGS="-DVIDAR_MSDA_TWAVES=$1 -DVIDAR_MSDA_CHUNK=$2" python -m vidar_amd.build > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== waves_per_wg=$1 chunk=$2"
  timeout 200 python tools/kbench.py msda 2>&1 | grep "binned=True" | cut -c1-120
done
python -m vidar_amd.build > /dev/null 2>&1
echo "default build restored"
