#!/bin/bash
# Counters of the blocked pair form of the psi2 term at config 5: two passes of SQ counters for the main kernel, then FETCH_SIZE
# and WRITE_SIZE in passes of their own for the record kernel and the main kernel of a Q = 40 term.
#   usage: tools/pmc_pairblk.sh OUTNAME       -> gpurun_out/OUTNAME/{counters40.json, traffic40.json}
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pmc_pairblk}
mkdir -p $O
export RGP_PSI2_PAIR=1
B="python3 bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- $B > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/b -- $B > $O/b.log 2>&1 || { tail -5 $O/b.log; exit 1; }
python tools/pmc_summary.py $O/counters40.json "psi2_pairblk_kernel<40" $O/a $O/b > /dev/null || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/c -- $B > $O/c.log 2>&1 || { tail -5 $O/c.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/d -- $B > $O/d.log 2>&1 || { tail -5 $O/d.log; exit 1; }
python - $O <<'PY'
import json, subprocess, sys
O = sys.argv[1]
tot = {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0}
per = {}
for k in ('psi2_pairblk_prep_kernel<40', 'psi2_pairblk_kernel<40'):
    r = json.loads(subprocess.check_output([sys.executable, 'tools/pmc_summary.py', O + '/tmp.json', k, O + '/c', O + '/d']).decode().strip().splitlines()[-1])
    per[k + '>'] = {n: r[n]['avg_per_launch_kb'] for n in tot}
    for n in tot:
        tot[n] += r[n]['avg_per_launch_kb']
json.dump({'FETCH_SIZE': {'avg_per_launch_kb': tot['FETCH_SIZE']}, 'WRITE_SIZE': {'avg_per_launch_kb': tot['WRITE_SIZE']},
           'per_kernel_kb': per, 'note': 'sum over the record kernel and the main kernel of one Q = 40 psi2 term of the blocked pair form (N = 524128, M = 500)'},
          open(O + '/traffic40.json', 'w'), indent=1)
print(json.dumps(per))
PY
cat $O/counters40.json
