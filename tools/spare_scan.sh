#!/bin/bash
# steps/s of short SVI steps (4 and 8 sequences) with the pair form forced, against the number of CUs its grid leaves free
for mb in ${MBS:-4 8}; do for sp in ${SPS:-6 16 26 36 46}; do
  RGP_PSI2_PAIR=2 RGP_PSI2_PAIR_SPARE=$sp timeout -k 10 200 python bench.py --n-total 32 --minibatch $mb --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null > /tmp/rs.json || exit 1
  python -c "import json; d=json.loads(open('/tmp/rs.json').readline()); print('minibatch $mb  free CUs $sp  %.2f steps/s' % d['value'])"
done; done
