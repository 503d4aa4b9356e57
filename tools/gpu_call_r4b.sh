#!/bin/bash
# Round 4, GEMM iteration call: parity tests, kernel timings, counters of csrc/gemm_mfma.hip
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/r4b
mkdir -p $out
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
stamp "gemm parity"
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -s 2>&1 | grep -v "^$" | tail -12 | tee $out/gemm_tests.log
stamp "gemm kbench"
timeout 300 python tools/kbench.py gemm 2>&1 | tee $out/kbench_gemm.log | cut -c1-330
stamp "gemm modes through the model"
timeout 900 python -m pytest tests/test_gemm_modes_gpu.py -q 2>&1 | tail -12 | tee $out/gemm_modes_tests.log
stamp "counters"
bash tools/pmc_pass.sh $out/pmc_gemm "SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_LDS,SQ_ACTIVE_INST_VMEM,SQ_WAIT_INST_LDS,GRBM_GUI_ACTIVE TA_BUSY_avr,TCP_TCC_READ_REQ,TCC_HIT,TCC_MISS FETCH_SIZE WRITE_SIZE" python tools/kbench.py gemm_pmc > $out/pmc_gemm.log 2>&1
tail -60 $out/pmc_gemm.log | cut -c1-200
stamp "done"
