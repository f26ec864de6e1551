#!/bin/bash
# HBM traffic of the wide-window path's large matrix products (FETCH_SIZE and WRITE_SIZE in separate passes; run on the GPU box
# from the repo root).  Summaries per product: T = (c o W) ZZ2^T is the <false, true> instance, C = (c o W)^T d2 the <true, false> one.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pw
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pw/$c
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pw/$c -- python3 bench.py --wide --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pw/$c.log 2>&1 || { tail -5 gpurun_out/pw/$c.log; exit 1; }
done
python tools/pmc_summary.py gpurun_out/pw/T_traffic.json 'gemm_f64_large_kernel<false, true>' gpurun_out/pw/FETCH_SIZE gpurun_out/pw/WRITE_SIZE
python tools/pmc_summary.py gpurun_out/pw/C_traffic.json 'gemm_f64_large_kernel<true, false>' gpurun_out/pw/FETCH_SIZE gpurun_out/pw/WRITE_SIZE
python tools/pmc_summary.py gpurun_out/pw/exp_traffic.json 'wq_exp_kernel' gpurun_out/pw/FETCH_SIZE gpurun_out/pw/WRITE_SIZE
