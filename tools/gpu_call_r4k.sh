#!/bin/bash
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4k
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "tests of the changed kernels"
timeout 600 python -m pytest tests/test_norm_fuse_gpu.py tests/test_gemm_gpu.py -q -x 2>&1 | tail -3 | tee $out/tests.log
stamp "private copies 8 / 16 / 32"
for v in default copies16 copies32; do
  cp vidar_amd/_variants/$v.so vidar_amd/libvidar_hip.so
  echo "== $v"; timeout 300 python tools/kbench.py ray lr 2>&1 | grep "_bwd" | cut -c1-120
done | tee $out/copies.log
cp vidar_amd/_variants/default.so vidar_amd/libvidar_hip.so
stamp "whole step"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" --op-table 2> $out/step.optable | tail -1 > $out/step.json
python -c "
import json; d = json.loads(open('$out/step.json').read()); print('step', round(d['ms_per_step'], 2), 'ms')"
stamp "rocprof of the step: colsum / slab reduce"
rm -rf /tmp/prof_step
( cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_step -o run -- python $OLDPWD/bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" ) > $out/bench_traced.json 2> $out/bench_traced.err
db=$(find /tmp/prof_step -name "*.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py $db --steps 10 > $out/step_kernel_summary.txt 2>&1
grep "colsum\|slab_reduce\|gemm_mfma\|^#" $out/step_kernel_summary.txt | cut -c1-150
stamp "done"
