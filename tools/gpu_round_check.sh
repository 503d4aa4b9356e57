#!/bin/bash
# What the driver runs at round end, in one call: the -m gpu suite, smoke(), the bench command (+ its kernel trace).
#     gpurun --timeout 1500 -- 'bash tools/gpu_round_check.sh [tag]'
# Everything lands in gpurun_out/<tag>/ ; copy what is kept into profiles/.
set -u
cd "$(dirname "$0")/.."
tag=${1:-round_check}
out=gpurun_out/$tag
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "full GPU suite"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $out/gpu_suite.log
stamp "smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.log
stamp "bench (driver command), hard limit 480 s"
timeout 480 python bench.py --gpus 1 --steps 20 --warmup 5 --op-table > $out/bench.json 2> $out/optable.txt
echo "rc $?"
grep "^\[bench\]" $out/optable.txt
python - $out/bench.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no json:", e); raise SystemExit
print("main:", round(d["ms_per_step"], 2), "ms/step", round(d["value"], 3), "samples/s peak_mem", d.get("peak_mem_gb"), "gemm", d.get("gemm"))
r = d["roofline"]; print("roofline:", r["kernel"], round(r["achieved"], 1), "GB/s frac", round(r["frac"], 4), "traffic", r["traffic"], "avg_ms", round(r["avg_ms"], 4))
for c in d.get("configs", []):
    print("  ", c.get("config"), "spg", c.get("samples_per_gpu"), c.get("gemm"), c.get("error") or c.get("skipped") or (round(c["ms_per_step"], 1), "ms", round(c["value"], 3), "samples/s", "peak", c.get("peak_mem_gb")))
PY
stamp "kernel trace of the bench command (timed region)"
rm -rf /tmp/prof_step
( cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_step -o run -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" ) > $out/bench_traced.json 2> $out/bench_traced.err
db=$(find /tmp/prof_step -name "*.db" | head -1)
[ -n "$db" ] && python tools/prof_summary.py $db --steps 20 > $out/step_kernel_summary.txt 2>&1
tail -7 $out/step_kernel_summary.txt
stamp "done"
