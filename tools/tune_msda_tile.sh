#!/bin/bash
# Tuning sweep of the msda_bwd accumulate kernel's compile-time constants: rebuilds libvidar_hip.so per variant
# and times `tools/kbench.py msda`.  Run on a GPU box; restores the default build.
#   VIDAR_MSDA_ACC 1 = register window (tile edge 4), 0 = private LDS window per wave (tile edge 8 or 4)
set -u
cd "$(dirname "$0")/.."
for v in "1 2 1024" "0 3 1024" "0 2 1024" "1 2 512" "1 2 4096"; do
  set -- $v
  VIDAR_EXTRA_HIPCC_FLAGS="-DVIDAR_MSDA_ACC=$1 -DVIDAR_MSDA_TILE_SHIFT=$2 -DVIDAR_MSDA_CHUNK=$3" python -m vidar_amd.build > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== acc=$1 tile_shift=$2 chunk=$3"
  timeout 200 python tools/kbench.py msda 2>&1 | grep "binned=True" | cut -c1-120
done
python -m vidar_amd.build > /dev/null 2>&1
echo "default build restored"
