#!/bin/bash
# headline record only (no extras, no CPU leg) + the kernel trace of its timed region.
#     gpurun --timeout 900 -- 'bash tools/gpu_quick_trace.sh [tag] [extra bench flags]'
set -u
cd "$(dirname "$0")/.."
tag=${1:-quick}; shift || true
out=gpurun_out/$tag; mkdir -p $out
rm -rf /tmp/prof_step
( cd /tmp && TMPDIR=/tmp timeout 500 rocprofv3 --kernel-trace -d /tmp/prof_step -o run -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" --op-table "$@" ) > $out/bench_traced.json 2> $out/bench_traced.err
grep "^\[bench\]" $out/bench_traced.err
db=$(find /tmp/prof_step -name "*.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py $db --steps 20 > $out/step_kernel_summary.txt 2>&1
tail -7 $out/step_kernel_summary.txt
