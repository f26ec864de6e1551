#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc4
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc4/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc4/a.log 2>&1 || { tail -5 gpurun_out/pmc4/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --kernel-trace --output-format csv -d gpurun_out/pmc4/b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc4/b.log 2>&1 || { tail -5 gpurun_out/pmc4/b.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pmc4/counters40.json "psi2_rowsP_kernel<40" gpurun_out/pmc4/a gpurun_out/pmc4/b
