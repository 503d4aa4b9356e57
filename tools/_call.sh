mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_dropin_gpu.py tests/test_reference_golden_gpu.py tests/test_step_gpu.py -x -q 2>&1 | tail -8) > gpurun_out/r03_c_tests.log
timeout 200 python tools/kbench.py msda msda_coherent > gpurun_out/r03_c_kbench_acc1.log 2>&1
G1=SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_SCA
G2=SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VMEM,SQ_WAIT_INST_LDS,SQ_INST_CYCLES_SALU
G3=TA_BUSY_avr,TCP_TOTAL_CACHE_ACCESSES_sum,TCP_TCC_READ_REQ_sum,TCC_HIT_sum,TCC_MISS_sum,TCP_PENDING_STALL_CYCLES_sum,GRBM_GUI_ACTIVE
tools/pmc_pass.sh gpurun_out/r03_pmc_acc1 "$G1 $G2 $G3 FETCH_SIZE WRITE_SIZE" python tools/kbench.py msda_sca > gpurun_out/r03_pmc_acc1.log 2>&1
VIDAR_EXTRA_HIPCC_FLAGS="-DVIDAR_MSDA_ACC=0" python -m vidar_amd.build > /dev/null 2>&1
timeout 200 python tools/kbench.py msda msda_coherent > gpurun_out/r03_c_kbench_acc0.log 2>&1
tools/pmc_pass.sh gpurun_out/r03_pmc_acc0 "$G1 $G2 $G3" python tools/kbench.py msda_sca > gpurun_out/r03_pmc_acc0.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o run -- python $GRAFT_REPO_ROOT/tools/kbench.py msda > /tmp/kt.log 2>&1; f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cut -c1-400 "$f" | grep -v "at::native" > $GRAFT_REPO_ROOT/gpurun_out/r03_kbench_msda_kernel_stats_acc0.csv)
python -m vidar_amd.build > /dev/null 2>&1
cat gpurun_out/r03_c_tests.log gpurun_out/r03_c_kbench_acc1.log gpurun_out/r03_c_kbench_acc0.log
