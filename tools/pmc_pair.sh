#!/bin/bash
# SQ counters of the pair-form psi2 kernel at config 4 (two passes of at most 8 SQ counters); usage: tools/pmc_pair.sh OUTNAME
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmc_pair}
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/b.log 2>&1 || { tail -5 $O/b.log; exit 1; }
python tools/pmc_summary.py $O/counters40.json "psi2_pair_kernel<40" $O/a $O/b
