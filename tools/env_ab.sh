#!/bin/bash
# A/B of an environment switch inside one box: tools/env_ab.sh VAR "<values>" "<mb list>"   (value "-" = variable unset)
var=$1
for mb in ${3:-4 8 32}; do
  for val in $2; do
    for rep in 1 2; do
      if [ "$val" = "-" ]; then pre="env -u $var"; else pre="env $var=$val"; fi
      v=$($pre python bench.py --steps 40 --warmup 5 --no-cpu-baseline --n-total 32 --minibatch $mb 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('%.2f steps/s  %.3f ms/step  psi2 %.1f TF'%(j['value'], j['ms_per_step'], r['achieved']))")
      echo "mb=$mb $var=$val rep=$rep $v"
    done
  done
done
