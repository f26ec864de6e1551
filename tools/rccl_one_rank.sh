for mb in 16 8 4; do
  for fd in 0 1; do
    for b in layer single; do
      if [ $fd = 0 ] && [ $b = single ]; then continue; fi
      v=$(RGP_BENCH_FORCE_DIST=$fd timeout -k 10 200 python bench.py --no-cpu-baseline --bucket $b --n-total 32 --minibatch $mb --steps 100 --warmup 10 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.2f steps/s %.3f ms'%(j['value'], j['ms_per_step']))")
      echo "mb=$mb rccl_1rank=$fd bucket=$b $v"
    done
  done
done
