#!/bin/bash
# measurements of the wide-window reading of config 4: its GPU tests, kernel statistics and the bench line (run on the GPU box from the repo root)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mw
timeout -k 10 300 python -u -m pytest tests/test_gpu_wide_windows.py tests/test_gpu_model.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/mw/tests.log 2>&1 || { grep -v "^E    " gpurun_out/mw/tests.log | tail -20; exit 1; }
tail -2 gpurun_out/mw/tests.log
rm -rf gpurun_out/mw/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mw/prof -- python3 -u bench.py --wide --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/mw/prof.json 2> gpurun_out/mw/prof.err || { tail -5 gpurun_out/mw/prof.err; exit 1; }
if [ "$1" = "full" ]; then
  timeout -k 10 600 python3 -u bench.py --wide > gpurun_out/mw/bench.json 2> gpurun_out/mw/bench.err || { tail -5 gpurun_out/mw/bench.err; exit 1; }
fi
find gpurun_out/mw/prof -name '*kernel_stats.csv' | head -2
