#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_f16_dma.txt
: > $OUT
COMMON="--steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints --precision f16x3"
for rep in 1 2; do
for sched in 0 1 2 3; do
  for model in W S; do
    echo "## sched $sched model $model" >> $OUT
    PK_MI355_F16_DMA=$sched timeout -k 10 300 python bench.py --model $model $COMMON 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s alg' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved']))" >> $OUT
  done
done
done
PK_MI355_F16_DMA=3 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f16x3_full_path or f16x3_ragged" >> $OUT 2>&1
cat $OUT
