mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_msda_gpu.py tests/test_step_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/r03_i_tests.log
timeout 200 python tools/kbench.py msda > gpurun_out/r03_i_kbench_overlap.log 2>&1
(timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --op-table --no-cpu-baseline --no-kernel-rooflines --extra-configs "" > gpurun_out/r03_i_bench.json 2> gpurun_out/r03_i_optable.txt)
cat gpurun_out/r03_i_tests.log; grep "binned=True" gpurun_out/r03_i_kbench_overlap.log | cut -c1-120; cut -c1-250 gpurun_out/r03_i_bench.json; grep -E "msda" gpurun_out/r03_i_optable.txt
