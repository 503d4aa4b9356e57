set -u
mkdir -p gpurun_out/r06_msda_seg
timeout 900 python -m pytest tests/test_msda_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/r06_msda_seg/tests.log
for v in 0 1; do
  echo "== tile variant $v" | tee -a gpurun_out/r06_msda_seg/kbench.log
  VIDAR_MSDA_TILE_VARIANT=$v timeout 300 python tools/kbench.py msda msda_sca_coherent 2>&1 | grep -v "^{\"device" | tee -a gpurun_out/r06_msda_seg/kbench.log
done
