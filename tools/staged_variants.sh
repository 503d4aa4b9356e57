#!/bin/bash
# Compile-time variants that are STAGED in the source -- written and checked without a GPU (logic on the host where the
# code is host-compilable, device code of the default build proven unchanged by disassembly), default off -- and wait
# for one GPU run each: parity tests first, then the kernel timing next to the default build.  Run on a GPU box:
#     tools/staged_variants.sh            # every variant
#     tools/staged_variants.sh dvr_pipe   # one
# The default build is restored at the end.  A variant is promoted by making its macro the default in the source.
set -u
cd "$(dirname "$0")/.."
want="${1:-all}"

run() {   # name, hipcc flags, pytest selection, kbench target, grep pattern of the kbench lines
  local name="$1" flags="$2" tests="$3" kb="$4" pat="$5"
  [ "$want" = all ] || [ "$want" = "$name" ] || return 0
  echo "=================== $name   ($flags)"
  VIDAR_EXTRA_HIPCC_FLAGS="$flags" python -m vidar_amd.build > /dev/null 2>&1 || { echo "build failed"; return 0; }
  timeout 900 python -m pytest $tests -x -q -m gpu 2>&1 | tail -2
  timeout 300 python tools/kbench.py $kb 2>&1 | grep -i "$pat" | cut -c1-160
}

echo "=================== default build"
python -m vidar_amd.build > /dev/null 2>&1
timeout 300 python tools/kbench.py dvr dcn affine ray 2>&1 | grep -i "render\|im2col\|col2im\|affine\|ray_" | cut -c1-160

# dvr family: a sample's density is consumed one commit later (dvr_march.h) -- bit-identical arithmetic, the load gets a
# whole traversal step to arrive; expect the most at <= 1 wave per SIMD (30 k rays), where nothing else hides it
run dvr_pipe "-DVIDAR_DVR_PIPELINED_SIGMA" "tests/test_dvr_gpu.py tests/test_fullsize_parity_gpu.py tests/test_dropin_gpu.py" dvr "render"
# DCN im2col: 8 channels per thread = 52 VGPRs, no scalar spills, 8 waves per SIMD (16: 85 / 27 / 5)
run dcn_cp8 "-DVIDAR_DCN_CP=8" "tests/test_dcn_gpu.py" dcn "im2col"
run dcn_cp4 "-DVIDAR_DCN_CP=4" "tests/test_dcn_gpu.py" dcn "im2col"
# DCN col2im, offset / mask gradient: the channel loop keeps 3 loads in flight and waits 256 times in a row (0.144 ms
# per call = 256 x one memory latency); the loads of 4 / 8 channels issued together (74 / 128 VGPRs, 6 / 4 waves)
run dcn_coord4 "-DVIDAR_DCN_COORD_BATCH=4" "tests/test_dcn_gpu.py" dcn "col2im"
run dcn_coord8 "-DVIDAR_DCN_COORD_BATCH=8" "tests/test_dcn_gpu.py" dcn "col2im"
run dcn_segscan "-DVIDAR_DCN_SEGMENTED_SCAN=1" "tests/test_dcn_gpu.py" dcn "col2im"
# ray kernels of the head: leave the 512-waypoint loop after the run of live waypoints (60-140 of 512: on average 3.9 of
# the 16 backward passes and 2.6 of the 8 forward passes are executed, tests/test_ray_early_exit_cpu.py)
run ray_early "-DVIDAR_RAY_EARLY_EXIT=1" "tests/test_ray_ops_gpu.py tests/test_head_loss_gpu.py tests/test_step_gpu.py tests/test_reference_golden_gpu.py" ray "ray_"
# frozen BN + residual + ReLU: 2 / 4 float4 per thread with all loads issued first (5.5 TB/s today, 6.3 achievable)
run aa_ilp2 "-DVIDAR_AA_ILP=2" "tests/test_dcn_gpu.py" affine "affine"
run aa_ilp4 "-DVIDAR_AA_ILP=4" "tests/test_dcn_gpu.py" affine "affine"

python -m vidar_amd.build > /dev/null 2>&1
echo "default build restored"
