#!/bin/bash
# measurements of config 5 at full size: bench line and kernel statistics (run on the GPU box from the repo root)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m5
timeout -k 10 600 python bench.py --config 5 > gpurun_out/m5/bench.json 2> gpurun_out/m5/bench.err || { tail -5 gpurun_out/m5/bench.err; exit 1; }
cut -c1-300 gpurun_out/m5/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/m5/prof -- python3 bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/m5/prof.json 2> gpurun_out/m5/prof.err || { tail -5 gpurun_out/m5/prof.err; exit 1; }
find gpurun_out/m5/prof -name '*kernel_stats.csv' | head -2
