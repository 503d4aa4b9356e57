mkdir -p gpurun_out
tools/tune_msda_tile.sh > gpurun_out/r03_msda_tile_sweep4.log 2>&1
(timeout 300 python -m pytest tests/test_msda_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/r03_f_tests.log
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python $GRAFT_REPO_ROOT/tools/kbench.py msda > $GRAFT_REPO_ROOT/gpurun_out/r03_f_kbench_msda.log 2>&1; f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); grep -v "at::native" "$f" > $GRAFT_REPO_ROOT/gpurun_out/r03_kbench_msda_kernel_stats_f.csv)
cat gpurun_out/r03_msda_tile_sweep4.log gpurun_out/r03_f_tests.log
