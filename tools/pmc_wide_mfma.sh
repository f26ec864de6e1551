#!/bin/bash
# MFMA-pipe counters of the wide-window path's large matrix products (one --pmc pass; run on the GPU box from the repo root)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pwm && rm -rf gpurun_out/pwm/sq
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pwm/sq -- python3 bench.py --wide --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pwm/sq.log 2>&1 || { tail -5 gpurun_out/pwm/sq.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pwm/T_counters.json 'gemm_f64_large_kernel<false, true>' gpurun_out/pwm/sq
python tools/pmc_summary.py gpurun_out/pwm/C_counters.json 'gemm_f64_large_kernel<true, false>' gpurun_out/pwm/sq
