#!/bin/bash
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4l
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
short() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{\"op'): continue
    d = json.loads(l); print('   ', d['op'][:52].ljust(52), ' '.join(f'{k[:-3]}={v:.4f}' for k, v in d.items() if k.endswith('_ms') and k != 'min_HBM_ms' and 'lib' not in k))
"; }
for st in 2 1; do
  stamp "gemm parity, stages $st"
  VIDAR_GEMM_STAGES=$st timeout 600 python -m pytest tests/test_gemm_gpu.py -q -x 2>&1 | tail -2
  stamp "gemm kbench, stages $st"
  VIDAR_GEMM_STAGES=$st timeout 300 python tools/kbench.py gemm 2>&1 | tee $out/kbench_gemm_stages$st.log | short
done
stamp "colsum + step"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('step auto', round(d['ms_per_step'], 2), 'ms')"
VIDAR_GEMM_STAGES=2 VIDAR_GEMM=bf16x3 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('step bf16x3 stages 2', round(d['ms_per_step'], 2), 'ms')"
VIDAR_GEMM_STAGES=1 VIDAR_GEMM=bf16x3 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('step bf16x3 stages 1', round(d['ms_per_step'], 2), 'ms')"
stamp "done"
