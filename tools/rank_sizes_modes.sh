#!/bin/bash
# Per-rank sizes of the 1 / 2 / 4 / 8 GPU runs of config 4 on ONE GPU (32 / 16 / 8 / 4 sequences per step) with the sweep
# kernels (mode 0) and the pair form (mode 1) of the fused psi2 pass.
for mode in ${MODES:-0 1}; do
  for mb in 32 16 8 4; do
    RGP_PSI2_PAIR=$mode timeout -k 10 200 python bench.py --n-total 32 --minibatch $mb --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null > /tmp/rs.json || exit 1
    python - "$mode" "$mb" <<'PY'
import json, sys
d = json.loads(open('/tmp/rs.json').readline())
print('psi2 mode %s  minibatch %2s  %7.2f steps/s  %7.3f ms/step  psi2 %.2f TFLOP/s' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['achieved']))
PY
  done
done
