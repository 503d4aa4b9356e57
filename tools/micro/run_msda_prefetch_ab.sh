set -u
out=gpurun_out/r06_msda_prefetch; mkdir -p $out
timeout 900 python -m pytest tests/test_msda_gpu.py -x -q 2>&1 | tail -3 | tee $out/tests.log
for grp in 8 16 32; do
  echo "== cross-batch operand prefetch, group $grp" | tee -a $out/kbench.log
  VIDAR_EXTRA_HIPCC_ONLY=msda.hip VIDAR_EXTRA_HIPCC_FLAGS="-DVIDAR_MSDA_TILE_GROUP=$grp" python -m vidar_amd.build > /dev/null 2>&1
  timeout 300 python tools/kbench.py msda msda_sca_coherent 2>&1 | grep "bwd.*binned=True\|coherent" | tee -a $out/kbench.log
done
python -m vidar_amd.build > /dev/null 2>&1
