#!/bin/bash
# Compile-time variants that are STAGED in the source -- written and checked without a GPU (logic on the host where the
# code is host-compilable, device code of the default build proven unchanged by `tools/devcode.py head-diff`), default
# off -- and wait for one GPU run each: parity tests first, then the kernel timing next to the default build.
#     tools/staged_variants.sh prebuild     # on the CPU box: one libvidar_hip.so per variant into vidar_amd/_staged/
#                                           # (git-ignored, travels with gpurun: no compile time on the GPU box)
#     tools/staged_variants.sh              # on a GPU box: every variant   (tools/staged_variants.sh dvr_pipe: one)
# The default build is restored at the end.  A variant is promoted by making its macro the default in the source.
set -u
cd "$(dirname "$0")/.."
want="${1:-all}"
staged=vidar_amd/_staged
lib=vidar_amd/libvidar_hip.so

# name | source file the flags apply to | hipcc flags | pytest selection | kbench target | grep pattern of the kbench lines
variants() { cat <<'TABLE'
dvr_pipe|dvr_family.hip|-DVIDAR_DVR_PIPELINED_SIGMA|tests/test_dvr_gpu.py tests/test_fullsize_parity_gpu.py tests/test_dropin_gpu.py|dvr|render
dcn_cp8|dcn.hip|-DVIDAR_DCN_CP=8|tests/test_dcn_gpu.py|dcn|im2col
dcn_cp4|dcn.hip|-DVIDAR_DCN_CP=4|tests/test_dcn_gpu.py|dcn|im2col
dcn_nt|dcn.hip|-DVIDAR_DCN_NT_STORES=1|tests/test_dcn_gpu.py|dcn|im2col
dcn_cp8_nt|dcn.hip|-DVIDAR_DCN_CP=8 -DVIDAR_DCN_NT_STORES=1|tests/test_dcn_gpu.py|dcn|im2col
dcn_coord4|dcn.hip|-DVIDAR_DCN_COORD_BATCH=4|tests/test_dcn_gpu.py|dcn|col2im
dcn_coord8|dcn.hip|-DVIDAR_DCN_COORD_BATCH=8|tests/test_dcn_gpu.py|dcn|col2im
dcn_segscan|dcn.hip|-DVIDAR_DCN_SEGMENTED_SCAN=1|tests/test_dcn_gpu.py|dcn|col2im
ray_early|ray_march.hip|-DVIDAR_RAY_EARLY_EXIT=1|tests/test_ray_ops_gpu.py tests/test_head_loss_gpu.py tests/test_step_gpu.py tests/test_reference_golden_gpu.py|ray|ray_
ray_copies8|ray_march.hip|-DVIDAR_RAY_COPIES=8|tests/test_ray_ops_gpu.py tests/test_head_loss_gpu.py tests/test_step_gpu.py|ray|ray_
ray_early_copies8|ray_march.hip|-DVIDAR_RAY_EARLY_EXIT=1 -DVIDAR_RAY_COPIES=8|tests/test_ray_ops_gpu.py tests/test_head_loss_gpu.py|ray|ray_
lr_copies8|latent_render.hip|-DVIDAR_LR_COPIES=8|tests/test_latent_render_gpu.py tests/test_step_gpu.py|lr|lr_
msda_skip|msda.hip|-DVIDAR_MSDA_SKIP_DEAD=1|tests/test_msda_gpu.py tests/test_step_gpu.py|msda|msda
msda_nt|msda.hip|-DVIDAR_MSDA_NT_LOADS=1|tests/test_msda_gpu.py|msda msda_coherent|msda
aa_ilp2|affine_act.hip|-DVIDAR_AA_ILP=2|tests/test_dcn_gpu.py|affine|affine
aa_ilp4|affine_act.hip|-DVIDAR_AA_ILP=4|tests/test_dcn_gpu.py|affine|affine
TABLE
}
# what each one is:
#  dvr_pipe    a sample's density is consumed one commit later (dvr_march.h): bit-identical arithmetic, the load gets a
#              whole traversal step to arrive; expect the most at <= 1 wave per SIMD (30 k rays)
#  dcn_cp8/4   im2col channels per thread: 16 = 85 VGPRs, 27 scalar registers parked in VGPR lanes, 5 waves; 8 = 52 / 0 / 8
#  dcn_nt      im2col column stores with the non-temporal policy (the 71 MB input is fetched 4 x from HBM today: the
#              1.28 GB write stream evicts the planes the next taps re-read)
#  dcn_coord*  offset / mask gradient: the loads of 4 / 8 channels issued together (today: 3 loads, wait, 256 times)
#  dcn_segscan col2im reverse map: one scan workgroup per (image, tap) list instead of per image
#  ray_early   leave the 512-waypoint loops after the run of live waypoints (on average 3.9 of 16 passes are needed)
#  ray_copies8 ray_ce / ray_gumbel backward: 8 private copies of the gradient volume + a sum kernel (all rays of a frame start
#              at the sensor origin: ~190 hot addresses take ~5 M atomics per ray_ce_bwd launch); the kbench line includes the
#              memset + sum of the copies -- read the kernel's own time from a rocprofv3 --kernel-trace if it is close
#  lr_copies8  LatentRendering backward: 8 private copies of the gradient maps + a sum kernel -- 8 x fewer atomics per hot
#              address (all rays start at the BEV centre) for the same total: tells contention from atomic rate
#  msda_skip   MSDA forward / grad_loc kernels: an item without a live sample issues no corner loads (pays off together with
#              VIDAR_SCA_PAD_NAN=1, timed in tools/first_gpu_call.sh; alone it only adds the check: expect +-0)
#  msda_nt     MSDA gathers (forward, grad_loc): corner lines with the non-temporal cache policy -- the gathers miss the
#              vector L1 on almost every line, and 0.43 ms is what fill-then-deliver costs at 64 B/clk (delivery alone: 0.23)
#  aa_ilp*     frozen BN + residual + ReLU: 2 / 4 float4 per thread, all loads first

build_variant() {   # file, flags
  VIDAR_EXTRA_HIPCC_ONLY="$1" VIDAR_EXTRA_HIPCC_FLAGS="$2" python -m vidar_amd.build > /dev/null 2>&1
}

if [ "$want" = prebuild ]; then
  mkdir -p $staged
  python -m vidar_amd.build > /dev/null 2>&1 && cp $lib $staged/default.so
  variants | while IFS='|' read -r name file flags tests kb pat; do
    build_variant "$file" "$flags" && cp $lib $staged/$name.so && echo "prebuilt $name" || echo "build failed: $name"
  done
  python -m vidar_amd.build > /dev/null 2>&1
  cmp -s $lib $staged/default.so || cp $staged/default.so $lib
  ls -la $staged
  exit 0
fi

use() {   # name, file, flags -> installs the variant library
  if [ -f $staged/$1.so ]; then cp $staged/$1.so $lib; else build_variant "$2" "$3"; fi
}

echo "=================== default build"
if [ -f $staged/default.so ]; then cp $staged/default.so $lib; else python -m vidar_amd.build > /dev/null 2>&1; fi
timeout 300 python tools/kbench.py dvr dcn affine ray lr msda 2>&1 | grep -i "render\|im2col\|col2im\|affine\|ray_\|lr_\|msda" | cut -c1-160

variants | while IFS='|' read -r name file flags tests kb pat; do
  [ "$want" = all ] || [ "$want" = "$name" ] || continue
  echo "=================== $name   ($flags)"
  use "$name" "$file" "$flags" || { echo "build failed"; continue; }
  [ -n "${VIDAR_VARIANTS_NOTEST:-}" ] || VIDAR_STAGED=1 timeout 900 python -m pytest $tests -x -q -m gpu 2>&1 | tail -2
  timeout 300 python tools/kbench.py $kb 2>&1 | grep -i "$pat" | cut -c1-160
done

if [ -f $staged/default.so ]; then cp $staged/default.so $lib; else python -m vidar_amd.build > /dev/null 2>&1; fi
echo "default build restored"
