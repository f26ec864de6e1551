#!/bin/bash
# kernel statistics of a 4-sequence step (the per-rank size of an 8-GPU run) with the pair form forced (mode 2) or the sweep kernels (0)
cd /tmp && export TMPDIR=/tmp
mode=${1:-2}
RGP_PSI2_PAIR=$mode timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_small_$mode -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --n-total 32 --minibatch 4 --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_small_$mode.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_small_$mode.log; exit 1; }
