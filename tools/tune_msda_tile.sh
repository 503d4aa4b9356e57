#!/bin/bash
# Tuning sweep of the msda_bwd tile kernel's compile-time constants (samples per chunk, waves per workgroup):
# rebuilds libvidar_hip.so per variant and times `tools/kbench.py msda`.  Run on a GPU box; restores the default build.
set -u
cd "$(dirname "$0")/.."
for v in "1024 2" "512 2" "2048 2" "1024 4" "1024 1" "4096 2"; do
  set -- $v
  VIDAR_EXTRA_HIPCC_FLAGS="-DVIDAR_MSDA_CHUNK=$1 -DVIDAR_MSDA_TWAVES=$2" python -m vidar_amd.build > /dev/null 2>&1
  echo "== chunk=$1 waves_per_wg=$2"
  timeout 200 python tools/kbench.py msda 2>&1 | grep "binned=True" | cut -c1-110
done
python -m vidar_amd.build > /dev/null 2>&1
echo "default build restored"
