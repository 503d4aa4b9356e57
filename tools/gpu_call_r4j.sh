#!/bin/bash
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4j
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "gemm parity"
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_gemm_modes_gpu.py -q -x 2>&1 | grep -v "Warning\|warn\|^$\|^  " | tail -6 | tee $out/gemm_tests.log
short() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{\"op'): continue
    d = json.loads(l); print('   ', d['op'][:52].ljust(52), ' '.join(f'{k[:-3]}={v:.4f}' for k, v in d.items() if k.endswith('_ms') and k != 'min_HBM_ms'))
"; }
stamp "gemm kbench"
timeout 300 python tools/kbench.py gemm 2>&1 | tee $out/kbench_gemm.log | short
step() {
  local label="$1"; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" --op-table \
      2> $out/step_$label.optable | tail -1 > $out/step_$label.json
  python -c "
import sys, json
d = json.loads(open('$out/step_$label.json').read()); print('$label', round(d['ms_per_step'], 2), 'ms/step')"
  grep "gemm_" $out/step_$label.optable | cut -c1-120
}
stamp "whole-step"
{ step auto VIDAR_GEMM=auto; step f32 VIDAR_GEMM=f32; step bf16x3 VIDAR_GEMM=bf16x3; } 2>&1 | tee $out/step_ab.log
stamp "done"
