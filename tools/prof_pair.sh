#!/bin/bash
# kernel statistics of the benchmark with the pair-form psi2 pass (mode $1, default 1) under rocprofv3
cd /tmp && export TMPDIR=/tmp
mode=${1:-1}
RGP_PSI2_PAIR=$mode timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pair_$mode -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_pair_$mode.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_pair_$mode.log; exit 1; }
head -14 $GRAFT_REPO_ROOT/gpurun_out/prof_pair_$mode/*kernel_stats.csv | cut -c1-150
