#!/bin/bash
# A/B of several builds on a reduced config-5 step inside one box: tools/lib_ab5.sh "<lib suffixes>"
for lib in $1; do
  v=$(RGP_HIP_LIB=$PWD/rgp_amd/librgp_$lib.so python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --n-total 2 --minibatch 2 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('%.4f steps/s  %.1f ms/step  %s %.1f TF (%.1f ms per launch)'%(j['value'], j['ms_per_step'], r['kernel'][:20], r['achieved'], r['avg_launch_ms']))")
  echo "lib=$lib $v"
done
