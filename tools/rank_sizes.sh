#!/bin/bash
# Per-rank sizes of the 2 / 4 / 8 GPU runs of config 4 on ONE GPU (16 / 8 / 4 sequences per step), then the two all-reduce
# modes with two ranks sharing the GPU over gloo (CPU-staged buckets; RGP_BENCH_ONE_GPU is a debugging aid of bench.py).
#   bash tools/rank_sizes.sh > profiles/rX_rank_sizes.txt
set -e
for mb in 32 16 8 4; do
  timeout -k 10 200 python bench.py --n-total 32 --minibatch $mb --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('minibatch %d  %.2f steps/s  %.3f ms/step  median %.3f ms  psi2 %.2f TFLOP/s' % ($mb, d['value'], d['ms_per_step'], d['step_ms_main_stream']['median'], d['roofline']['achieved']))"
done
for b in layer single; do
  RGP_BENCH_ONE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --bucket $b --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | \
    python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('2 ranks on one GPU over gloo, bucket=$b  %.2f steps/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"
done
