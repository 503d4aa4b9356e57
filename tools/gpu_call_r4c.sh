#!/bin/bash
# Round 4, GEMM iteration call: parity + ablation timings of csrc/gemm_mfma.hip
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4c
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "gemm parity"
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -x 2>&1 | grep -v "^$" | tail -12 | tee $out/gemm_tests.log
stamp "gemm kbench"
timeout 300 python tools/kbench.py gemm 2>&1 | grep -v "^/opt" | tee $out/kbench_gemm.log | cut -c1-330
for a in 1 2 4 8 3 5 6 7; do
  stamp "ablate $a (1 no loads, 2 no mfma, 4 no stores, 8 no lds staging)"
  VIDAR_GEMM_ABLATE=$a timeout 300 python tools/kbench.py gemm 2>&1 | grep "value_proj SCA\|layer3 conv3\|layer3 dcn" | grep "fwd" | tee $out/kbench_gemm_ablate$a.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['op'][:46], {k: v for k, v in d.items() if k.endswith('_ms') and ('f32' in k or 'bf16' in k)})
"
done
stamp "gemm modes through the model"
timeout 900 python -m pytest tests/test_gemm_modes_gpu.py -q 2>&1 | grep -v "Warning\|warn\|^$\|^  " | tail -40 | cut -c1-250 | tee $out/gemm_modes_tests.log
stamp "done"
