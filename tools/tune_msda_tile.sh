#!/bin/bash
# Tuning sweep of the msda_bwd accumulate kernel's compile-time constants (tile edge, records per chunk): rebuilds
# libvidar_hip.so per variant and times `tools/kbench.py msda`.  Run on a GPU box; restores the default build.
# (The round-3 sweeps over the accumulation scheme itself -- shared ds_add window, register window, 8-byte lane map,
#  resident grids -- are recorded in profiles/r03_msda_tile_sweep_*.log; those variants were removed from the source.)
set -u
cd "$(dirname "$0")/.."
for v in "3 1024" "3 512" "3 2048" "2 1024"; do
  set -- $v
  VIDAR_EXTRA_HIPCC_ONLY=msda.hip VIDAR_EXTRA_HIPCC_FLAGS="-DVIDAR_MSDA_TILE_SHIFT=$1 -DVIDAR_MSDA_CHUNK=$2" python -m vidar_amd.build > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "== tile_shift=$1 chunk=$2"
  timeout 200 python tools/kbench.py msda 2>&1 | grep "binned=True" | cut -c1-120
done
python -m vidar_amd.build > /dev/null 2>&1
echo "default build restored"
