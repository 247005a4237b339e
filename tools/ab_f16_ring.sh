#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_f16_ring.txt
: > $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f16 or wide or determinism or wider" >> $OUT 2>&1 || { tail -30 $OUT; exit 1; }
COMMON="--steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints --precision f16x3"
for rep in 1 2; do
for ring in 5 4; do
  for model in W S; do
    echo "## ring $ring model $model" >> $OUT
    PK_MI355_F16_RING=$ring timeout -k 10 300 python bench.py --model $model $COMMON 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s alg' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved']))" >> $OUT
  done
done
done
cat $OUT
