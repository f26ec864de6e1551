#!/bin/bash
# kernel statistics of the config-5 benchmark under rocprofv3 (RGP_PSI2_PAIR mode $1, default 1)
cd /tmp && export TMPDIR=/tmp
mode=${1:-1}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg5_$mode
RGP_PSI2_PAIR=$mode timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
head -16 $out/*kernel_stats.csv | cut -c1-160
