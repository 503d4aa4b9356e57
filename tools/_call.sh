mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_ray_ops_gpu.py tests/test_head_loss_gpu.py tests/test_msda_gpu.py tests/test_step_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/r03_g_tests.log
timeout 200 python tools/kbench.py ray msda msda_coherent > gpurun_out/r03_g_kbench.log 2>&1
tools/pmc_pass.sh gpurun_out/r03_pmc_msda_sca "FETCH_SIZE WRITE_SIZE TA_BUSY_avr,TCP_TOTAL_CACHE_ACCESSES_sum,TCP_TCC_READ_REQ_sum,TCC_HIT_sum,TCC_MISS_sum,TCP_PENDING_STALL_CYCLES_sum,GRBM_GUI_ACTIVE SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS" python tools/kbench.py msda_sca > gpurun_out/r03_pmc_msda_sca.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python $GRAFT_REPO_ROOT/tools/kbench.py msda > /dev/null 2>&1; f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); grep -v "at::native" "$f" > $GRAFT_REPO_ROOT/gpurun_out/r03_kbench_msda_kernel_stats_final.csv)
timeout 600 python tools/pretune_gemms.py --lo 4096 --hi 16384 --cams 6 8 12 --out gpurun_out/tunableop_sca_lengths.csv > gpurun_out/r03_pretune.log 2>&1
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --op-table > gpurun_out/r03_g_bench_driver_like.json 2> gpurun_out/r03_g_optable.txt)
cat gpurun_out/r03_g_tests.log; grep -v Warn gpurun_out/r03_g_kbench.log | cut -c1-120 | head -30; tail -2 gpurun_out/r03_pretune.log; cut -c1-300 gpurun_out/r03_g_bench_driver_like.json; grep -v Warn gpurun_out/r03_g_optable.txt | head -20
