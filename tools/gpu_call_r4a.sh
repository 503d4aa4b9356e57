#!/bin/bash
# Round 4, first GPU call (one gpurun call, ~20 GPU-minutes):
#   1. csrc/gemm_mfma.hip: parity tests, kernel timings next to the library GEMMs, whole-step A/B of the three GEMM modes
#   2. every staged compile-time variant of round 3: KERNEL TIMING ONLY next to the default build (the parity suites of the
#      winners run when they become the default) -- decides which macros are promoted and which are deleted
#   3. whole-step A/Bs of the staged run-time switches
#   4. HBM traffic counters of the MSDA kernels on coherent reference points
# Everything lands in gpurun_out/r4a/ ; what is kept is copied into profiles/.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4a
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }

stamp "gemm parity"
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -15 | tee $out/gemm_tests.log
stamp "gemm kbench"
timeout 300 python tools/kbench.py gemm 2>&1 | tee $out/kbench_gemm.log | cut -c1-260
stamp "gemm modes through the model"
timeout 900 python -m pytest tests/test_gemm_modes_gpu.py -x -q 2>&1 | tail -8 | tee $out/gemm_modes_tests.log

step() {   # label, library (default | msda_skip), environment assignments / bench flags -> one short bench run
  local label="$1" v="$2"; shift 2
  [ -f vidar_amd/_staged/$v.so ] && cp vidar_amd/_staged/$v.so vidar_amd/libvidar_hip.so
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
  grep "gemm_\|msda_\|affine_act\|stem_" $out/step_$label.optable | cut -c1-120
}
stamp "whole-step A/B"
{
  step baseline default VIDAR_GEMM=lib
  step gemm_f32 default VIDAR_GEMM=f32
  step gemm_bf16x3 default VIDAR_GEMM=bf16x3
  step fused_stem default VIDAR_FUSED_STEM=1
  step msda_skip+sca_pad_nan msda_skip VIDAR_SCA_PAD_NAN=1
  step fused_adamw default VIDAR_FUSED_ADAMW=1
} 2>&1 | tee $out/step_ab.log
[ -f vidar_amd/_staged/default.so ] && cp vidar_amd/_staged/default.so vidar_amd/libvidar_hip.so

stamp "staged variants, kernel timing only"
VIDAR_VARIANTS_NOTEST=1 bash tools/staged_variants.sh > $out/staged_variants.log 2>&1
tail -100 $out/staged_variants.log | cut -c1-200

stamp "private-copies experiments: the kernels' own times"
for v in default ray_copies8 lr_copies8; do
  [ -f vidar_amd/_staged/$v.so ] || continue
  cp vidar_amd/_staged/$v.so vidar_amd/libvidar_hip.so
  rm -rf /tmp/prof_$v
  ( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o run -- \
      python $OLDPWD/tools/kbench.py ray lr ) > $out/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; [ -n "$f" ] && grep -i "ray_\|lr_\|sum_copies\|fillBuffer" "$f" | cut -d, -f1-4 | cut -c1-150
done | tee $out/copies_kernel_times.log
[ -f vidar_amd/_staged/default.so ] && cp vidar_amd/_staged/default.so vidar_amd/libvidar_hip.so

stamp "staged tests that need the default library"
VIDAR_STAGED=1 timeout 300 python -m pytest tests/test_dcn_gpu.py tests/test_msda_gpu.py -q -m gpu -k "fused_stem or nan_locations_between" 2>&1 | tail -3 | tee $out/staged_tests_default_lib.log

stamp "PMC traffic of the MSDA kernels on coherent reference points"
bash tools/pmc_pass.sh $out/pmc_msda_coherent "FETCH_SIZE WRITE_SIZE TCC_HIT,TCC_MISS" python tools/kbench.py msda_coherent > $out/pmc_msda_coherent.log 2>&1
tail -30 $out/pmc_msda_coherent.log | cut -c1-200
stamp "done"
