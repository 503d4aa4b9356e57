#!/bin/bash
# whole-step A/B of the three GEMM modes + the model-level parity of the MFMA modes + the full GPU suite
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4e
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "gemm modes through the model"
timeout 900 python -m pytest tests/test_gemm_modes_gpu.py -q 2>&1 | grep -v "Warning\|warn\|^$\|^  " | tail -30 | cut -c1-250 | tee $out/gemm_modes_tests.log
step() {
  local label="$1"; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" --op-table \
      2> $out/step_$label.optable | tail -1 > $out/step_$label.json
  python - "$label" $out/step_$label.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[2]).read())
    print(sys.argv[1], round(d["ms_per_step"], 2), "ms/step  peak_mem_gb", d.get("peak_mem_gb"), " gemm", d.get("gemm"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep "gemm_\|affine_act" $out/step_$label.optable | cut -c1-120
}
stamp "whole-step A/B"
{
  step lib VIDAR_GEMM=lib
  step auto VIDAR_GEMM=auto
  step auto_nofuse VIDAR_GEMM=auto VIDAR_AUTO_FUSE_RES=0

  step bf16x3 VIDAR_GEMM=bf16x3
} 2>&1 | tee $out/step_ab.log
stamp "full GPU suite"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $out/gpu_suite.log
stamp "done"
