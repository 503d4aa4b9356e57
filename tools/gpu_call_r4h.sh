#!/bin/bash
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4h
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "msda tests, both item orders"
timeout 600 python -m pytest tests/test_msda_gpu.py -q -x 2>&1 | tail -4 | tee $out/msda_tests.log
for o in 0 1; do
  stamp "kbench msda, item order $o"
  VIDAR_MSDA_ITEM_ORDER=$o timeout 300 python tools/kbench.py msda msda_coherent 2>&1 | grep "msda_fwd\|binned=True" | cut -c1-150 | tee $out/kbench_msda_order$o.log
done
stamp "rocprofv3 kernel times, both orders"
for o in 0 1; do
  rm -rf /tmp/prof_o$o
  ( cd /tmp && VIDAR_MSDA_ITEM_ORDER=$o TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_o$o -o run -- python $OLDPWD/tools/kbench.py msda_sca ) > $out/prof_o$o.log 2>&1
  f=$(find /tmp/prof_o$o -name "*kernel_stats.csv" | head -1)
  echo "== order $o"; [ -n "$f" ] && grep -i "msda" "$f" | cut -d, -f1-4 | sed 's/(anonymous namespace):://' | cut -c1-120
done | tee $out/kernel_times.log
stamp "PMC both orders (FETCH_SIZE, hits)"
for o in 0 1; do
  VIDAR_MSDA_ITEM_ORDER=$o bash tools/pmc_pass.sh $out/pmc_order$o "FETCH_SIZE TCC_HIT,TCC_MISS" python tools/kbench.py msda_sca > $out/pmc_order$o.log 2>&1
  echo "== order $o"; grep "msda_fwd\|msda_bwd_locw" $out/pmc_order$o/*.csv | cut -c1-160
done
stamp "done"
