#!/bin/bash
# does the weight prescale itself cost clock?  same library, prescale on / off, W and S, twice
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4b_prescale_ab.txt
: > $OUT
for rep in 1 2; do
for np in 0 1; do
  for model in W S; do
    echo "## PK_MI355_NO_PRESCALE=$np model $model" >> $OUT
    PK_MI355_NO_PRESCALE=$np timeout -k 10 300 python bench.py --model $model --precision f16x3 --steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s alg  stages %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved'], {k: round(v, 3) for k, v in d['stage_ms_per_step'].items()}))" >> $OUT
  done
done
done
cat $OUT
