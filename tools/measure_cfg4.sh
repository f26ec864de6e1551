#!/bin/bash
# measurements of config 4: GPU tests, bench line, kernel statistics, HBM traffic of the psi2 kernel (run on the GPU box from the repo root)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m4
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/m4/tests.log 2>&1 || { tail -20 gpurun_out/m4/tests.log; exit 1; }
tail -2 gpurun_out/m4/tests.log
timeout -k 10 400 python bench.py > gpurun_out/m4/bench.json 2> gpurun_out/m4/bench.err || { tail -5 gpurun_out/m4/bench.err; exit 1; }
cut -c1-300 gpurun_out/m4/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/m4/prof -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/m4/prof.json 2> gpurun_out/m4/prof.err || { tail -5 gpurun_out/m4/prof.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/m4/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/m4/pmc_fetch.log 2>&1 || { tail -5 gpurun_out/m4/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/m4/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/m4/pmc_write.log 2>&1 || { tail -5 gpurun_out/m4/pmc_write.log; exit 1; }
python tools/pmc_summary.py gpurun_out/m4/traffic.json psi2_rowsP gpurun_out/m4/pmc_fetch gpurun_out/m4/pmc_write
find gpurun_out/m4/prof -name '*kernel_stats.csv' | head -2
