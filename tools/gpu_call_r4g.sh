#!/bin/bash
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4g
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "bench (driver command), hard limit 420 s"
timeout 420 python bench.py --gpus 1 --steps 20 --warmup 5 --op-table > $out/bench.json 2> $out/optable.txt
echo "rc $?"
grep "^\[bench\]" $out/optable.txt
python - $out/bench.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no json:", e); raise SystemExit
print("main:", round(d["ms_per_step"], 2), "ms/step", round(d["value"], 3), "samples/s peak_mem", d.get("peak_mem_gb"), "gemm", d.get("gemm"))
print("roofline:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline"].items()})
for c in d.get("configs", []):
    print("  ", c.get("config"), "spg", c.get("samples_per_gpu"), c.get("gemm"), c.get("error") or c.get("skipped") or (round(c["ms_per_step"], 1), "ms", round(c["value"], 3), "samples/s", "peak", c.get("peak_mem_gb")))
cb = d.get("cpu_baseline", {})
print("cpu_baseline:", cb.get("value"), cb.get("cores"), cb.get("all_cores_column"), cb.get("sample", "")[:80])
for r in cb.get("ops", []):
    print("   ", r["op"], r["cpu_ms"], r.get("cpu_ms_all_cores"), r.get("gpu_ms"), r.get("speedup"), r.get("speedup_all_cores"))
PY
grep -v "^\[bench\]" $out/optable.txt | head -32
stamp "done"
