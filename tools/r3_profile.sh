cd $GRAFT_REPO_ROOT
# (the matrix-pipe probes, tools/mfma_*_probe.hip, are built with hipcc --offload-arch=gfx950 and run on their own)
timeout -k 10 400 python3 bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err || { tail -5 gpurun_out/r3_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_stats -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3_stats.log 2>&1 || exit 1
