#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_plain_f16.txt
: > $OUT
COMMON="--steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints --precision f16"
for rep in 1 2; do
for shape in 16 32; do
  for model in W S; do
    echo "## plain f16, shape $shape model $model" >> $OUT
    PK_MI355_F16_SHAPE=$shape timeout -k 10 300 python bench.py --model $model $COMMON 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s alg' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved']))" >> $OUT
  done
done
done
cat $OUT
