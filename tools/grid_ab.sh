#!/bin/bash
# A/B of the psi2 grid size inside one box: tools/grid_ab.sh "<mb list>" "<grid list>"
for mb in ${1:-4 8 32}; do
  for g in ${2:-0 248}; do
    for rep in 1 2; do
      v=$(RGP_PSI2_GRID=$g python bench.py --steps 40 --warmup 5 --no-cpu-baseline --n-total 32 --minibatch $mb 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('%.2f steps/s  %.3f ms/step  psi2 %.1f TF frac %.3f'%(j['value'], j['ms_per_step'], r['achieved'], r['frac']))")
      echo "mb=$mb grid=$g rep=$rep $v"
    done
  done
done
