#!/bin/bash
# A/B of several builds inside one box: tools/lib_ab.sh "<lib suffixes>" "<mb list>" [env assignments]
# (rgp_amd/librgp_<suffix>.so; an optional third argument such as RGP_PSI2_NO_SKEW=1 is put in front of the command)
for mb in ${2:-32}; do
  for lib in $1; do
    for rep in 1 2; do
      v=$(env $3 RGP_HIP_LIB=$PWD/rgp_amd/librgp_$lib.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --n-total 32 --minibatch $mb 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('%.2f steps/s  %.3f ms/step  psi2 %.1f TF'%(j['value'], j['ms_per_step'], r['achieved']))")
      echo "mb=$mb lib=$lib $3 rep=$rep $v"
    done
  done
done
