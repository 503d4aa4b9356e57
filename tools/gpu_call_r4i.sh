#!/bin/bash
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4i
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
step() {
  local label="$1"; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" --op-table \
      2> $out/step_$label.optable | tail -1 > $out/step_$label.json
  python -c "
import sys, json
d = json.loads(open('$out/step_$label.json').read()); print('$label', round(d['ms_per_step'], 2), 'ms/step')"
  grep "msda_" $out/step_$label.optable | cut -c1-120
}
stamp "whole-step A/B of the MSDA item order"
{ step banded VIDAR_MSDA_ITEM_ORDER=0; step head_major VIDAR_MSDA_ITEM_ORDER=1; step banded2 VIDAR_MSDA_ITEM_ORDER=0; step head_major2 VIDAR_MSDA_ITEM_ORDER=1; } 2>&1 | tee $out/step_ab.log
stamp "PMC traffic, coherent SCA queries, head-major"
bash tools/pmc_pass.sh $out/pmc_msda_sca_coherent "FETCH_SIZE WRITE_SIZE TCC_HIT,TCC_MISS,TA_BUSY_avr,GRBM_GUI_ACTIVE" python tools/kbench.py msda_sca_coherent > $out/pmc_msda_sca_coherent.log 2>&1
tail -40 $out/pmc_msda_sca_coherent.log | grep msda | cut -c1-160
stamp "kernel trace of the driver's bench command (timed region)"
rm -rf /tmp/prof_step
( cd /tmp && TMPDIR=/tmp timeout 500 rocprofv3 --kernel-trace -d /tmp/prof_step -o run -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" ) > $out/bench_traced.json 2> $out/bench_traced.err
db=$(find /tmp/prof_step -name "*.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py $db --steps 20 > $out/step_kernel_summary.txt 2>&1
tail -12 $out/step_kernel_summary.txt
tail -1 $out/bench_traced.json | cut -c1-200
stamp "done"
