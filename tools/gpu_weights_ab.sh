#!/bin/bash
# The driver's bench command under the default (trained_like) weights + kernel traces of the timed region under BOTH
# weight states.     gpurun --timeout 1500 -- 'bash tools/gpu_weights_ab.sh [tag]'
set -u
cd "$(dirname "$0")/.."
tag=${1:-weights_ab}
out=gpurun_out/$tag
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "bench (driver command), hard limit 600 s"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --op-table > $out/bench.json 2> $out/optable.txt
echo "rc $?"
grep "^\[bench\]" $out/optable.txt
for w in trained_like init; do
  stamp "kernel trace of the timed region, weights=$w"
  rm -rf /tmp/prof_step
  ( cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_step -o run -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --weights $w --no-cpu-baseline --no-kernel-rooflines --extra-configs "" --op-table ) > $out/bench_traced_$w.json 2> $out/bench_traced_$w.err
  db=$(find /tmp/prof_step -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_summary.py $db --steps 20 > $out/step_kernel_summary_$w.txt 2>&1
  tail -5 $out/step_kernel_summary_$w.txt
done
stamp "done"
