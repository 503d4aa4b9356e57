#!/bin/bash
# Tuning sweep of the msda_bwd accumulate kernel's compile-time constants: rebuilds libvidar_hip.so per variant
# and times `tools/kbench.py msda`.  Run on a GPU box; restores the default build.
#   shared window (ds_add_f32) vs private windows (read-modify-write), tile edge 4 / 8 / 16, samples per chunk
set -u
cd "$(dirname "$0")/.."
for v in "1 3 4096" "0 3 1024" "1 4 4096" "1 2 4096" "1 3 2048" "1 3 8192" "1 4 8192"; do
  set -- $v
  VIDAR_EXTRA_HIPCC_FLAGS="-DVIDAR_MSDA_SHARED_WIN=$1 -DVIDAR_MSDA_TILE_SHIFT=$2 -DVIDAR_MSDA_CHUNK=$3" python -m vidar_amd.build > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== shared_win=$1 tile_shift=$2 chunk=$3"
  timeout 200 python tools/kbench.py msda 2>&1 | grep "binned=True" | cut -c1-120
done
python -m vidar_amd.build > /dev/null 2>&1
echo "default build restored"
