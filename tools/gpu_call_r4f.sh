#!/bin/bash
# full GPU suite + smoke + the driver's bench command on the cleaned-up default build
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4f
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "full GPU suite"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $out/gpu_suite.log
stamp "smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke.log
stamp "bench (driver command)"
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --op-table > $out/bench.json 2> $out/optable.txt
python - $out/bench.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("main:", round(d["ms_per_step"], 2), "ms/step", round(d["value"], 3), "samples/s peak_mem", d.get("peak_mem_gb"), "gemm", d.get("gemm"))
print("roofline:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline"].items()})
for c in d.get("configs", []):
    print("  ", c.get("config"), "spg", c.get("samples_per_gpu"), c.get("gemm"), c.get("error") or (round(c["ms_per_step"], 1), "ms", round(c["value"], 3), "samples/s", "peak", c.get("peak_mem_gb")))
cb = d.get("cpu_baseline", {})
print("cpu_baseline:", cb.get("value"), cb.get("cores"), cb.get("all_cores_column"))
for r in cb.get("ops", []):
    print("   ", r["op"], r["cpu_ms"], r.get("cpu_ms_all_cores"), r.get("gpu_ms"), r.get("speedup"), r.get("speedup_all_cores"))
PY
head -30 $out/optable.txt
stamp "done"
