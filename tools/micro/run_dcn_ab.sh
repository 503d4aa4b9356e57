set -u
out=gpurun_out/r06_dcn_lds; mkdir -p $out
timeout 900 python -m pytest tests/test_dcn_gpu.py -x -q 2>&1 | tail -5 | tee $out/tests.log
for v in 0 1; do
  echo "== dcn variant $v" | tee -a $out/kbench.log
  VIDAR_DCN_VARIANT=$v timeout 300 python tools/kbench.py dcn 2>&1 | grep -v "^{\"device\|amdgpu.ids" | tee -a $out/kbench.log
done
