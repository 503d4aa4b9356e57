#!/bin/bash
# The measurements that were prepared without a GPU and are waiting for one:
#     bash tools/staged_variants.sh prebuild        # here, on the CPU box: the variant libraries travel with the snapshot
#     gpurun --timeout 1500 -- 'bash tools/first_gpu_call.sh'
# 1. every staged compile-time variant: parity tests, then kernel timing next to the default build
# 2. HBM traffic counters of the MSDA kernels on COHERENT reference points (the committed numbers are for random ones)
# 3. whole-step A/Bs of the run-time switches (fused stem, NaN-padded cross-attention slots, fused AdamW)
# In full this is about 30 GPU-minutes; `tools/staged_variants.sh <name>` runs one variant.
# Everything lands in gpurun_out/first_call/ ; copy what is kept into profiles/.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/first_call
mkdir -p $out
# the tests that were written without a GPU, first on the DEFAULT library (a failure here is a bug of the test)
VIDAR_STAGED=1 timeout 300 python -m pytest tests/test_msda_gpu.py -q -m gpu -k nan_locations_between 2>&1 | tail -3 | tee $out/staged_tests_default_lib.log
bash tools/staged_variants.sh > $out/staged_variants.log 2>&1
tail -60 $out/staged_variants.log
# 1b. the private-copies experiments: the kernels' own times (the kbench lines include the memset and the sum of the copies)
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
# 2b. the staged stem kernel (BN + ReLU + max-pool in one pass): its bit-exactness test, then the whole-step A/B
VIDAR_STAGED=1 timeout 300 python -m pytest tests/test_dcn_gpu.py -q -m gpu -k fused_stem 2>&1 | tail -2 | tee $out/fused_stem_test.log
timeout 300 python tools/kbench.py stem 2>&1 | grep stem | tee $out/kbench_stem.log
step() {   # label, library (default | msda_skip), environment assignments -> one short bench run, ms/step + the MSDA op rows
  local label="$1" v="$2"; shift 2
  [ -f vidar_amd/_staged/$v.so ] && cp vidar_amd/_staged/$v.so vidar_amd/libvidar_hip.so
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines --extra-configs "" --op-table \
      2> $out/step_$label.optable | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'], 2), 'ms/step')"
  grep "msda_\|affine_act_fwd\|stem_" $out/step_$label.optable | cut -c1-110
}
# 2c. padded SpatialCrossAttention slots with NaN anchors (skipped by the MSDA kernels): step parity first
VIDAR_SCA_PAD_NAN=1 timeout 600 python -m pytest tests/test_step_gpu.py tests/test_reference_golden_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee $out/sca_pad_nan_test.log
# 3. whole-step A/Bs against ONE baseline run (10 steps each, ~1 min per run)
{
  step baseline default VIDAR_NOOP=1
  step fused_stem default VIDAR_FUSED_STEM=1
  step stem_s2d default VIDAR_STEM_S2D=1
  step sca_pad_nan default VIDAR_SCA_PAD_NAN=1
  step msda_skip msda_skip VIDAR_NOOP=1
  step msda_skip+sca_pad_nan msda_skip VIDAR_SCA_PAD_NAN=1
  step fused_adamw default VIDAR_FUSED_ADAMW=1
} | tee $out/step_ab.log
[ -f vidar_amd/_staged/default.so ] && cp vidar_amd/_staged/default.so vidar_amd/libvidar_hip.so
bash tools/pmc_pass.sh $out/pmc_msda_coherent "FETCH_SIZE WRITE_SIZE TCC_HIT,TCC_MISS" python tools/kbench.py msda_coherent > $out/pmc_msda_coherent.log 2>&1
ls $out/pmc_msda_coherent 2>/dev/null && head -20 $out/pmc_msda_coherent/*.csv
