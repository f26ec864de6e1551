#!/bin/bash
# A/B of two builds inside one box: tools/ab.sh "<mb list>"   (rgp_amd/librgp_prev.so = build of the previous commit)
for mb in ${1:-4 32}; do
  for lib in prev hip; do
    for rep in 1 2; do
      v=$(RGP_HIP_LIB=$PWD/rgp_amd/librgp_$lib.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --n-total 32 --minibatch $mb 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('%.2f steps/s  %.3f ms/step  psi2 %.1f TF  elbo %.12e'%(j['value'], j['ms_per_step'], r['achieved'], j.get('elbo', float('nan'))))")
      echo "mb=$mb lib=$lib rep=$rep $v"
    done
  done
done
