set -u
out=gpurun_out/r06_dcn_sort; mkdir -p $out
timeout 600 python -m pytest tests/test_dcn_gpu.py -x -q 2>&1 | tail -3 | tee $out/tests.log
for v in 0 1; do
  echo "== sort bins $v" | tee -a $out/kbench.log
  VIDAR_DCN_SORT_BINS=$v timeout 300 python tools/kbench.py dcn 2>&1 | grep "col2im.*gather=True" | tee -a $out/kbench.log
done
