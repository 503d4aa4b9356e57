#!/bin/bash
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4d
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "gemm parity"
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -x 2>&1 | grep -v "^$" | tail -12 | tee $out/gemm_tests.log
short() { python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{\"op'): continue
    d = json.loads(l); print('   ', d['op'][:52].ljust(52), ' '.join(f'{k[:-3]}={v:.4f}' for k, v in d.items() if k.endswith('_ms') and k != 'min_HBM_ms'))
"; }
stamp "gemm kbench (persistent, cross-tile prefetch)"
timeout 300 python tools/kbench.py gemm 2>&1 | tee $out/kbench_gemm.log | short
stamp "no cross-tile prefetch (ablate 16)"
VIDAR_GEMM_ABLATE=16 timeout 300 python tools/kbench.py gemm 2>&1 | tee $out/kbench_gemm_noprefetch.log | short
stamp "one workgroup per tile (VIDAR_GEMM_PERSIST=0)"
VIDAR_GEMM_PERSIST=0 timeout 300 python tools/kbench.py gemm 2>&1 | tee $out/kbench_gemm_nopersist.log | short
stamp "done"
