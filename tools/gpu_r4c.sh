#!/bin/bash
# tail lane: parity + A/B (PK_MI355_TAIL_LANE=0/1, lite kernel on/off), f16x3, models S and W
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tail_lane or f16x3_ragged or full_size or wider" > gpurun_out/r4c_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4c_tests.log
tail -8 gpurun_out/r4c_tests.log
OUT=gpurun_out/r4c_tail_lane_ab.txt
: > $OUT
for rep in 1 2; do
for cfg in "0 1 131072" "1 0 131072" "1 1 131072" "1 1 65536" "1 1 32768"; do
  set -- $cfg
  for model in W S; do
    echo "## PK_MI355_TAIL_LANE=$1 PK_MI355_TAIL_LITE=$2 PK_MI355_CHUNK=$3 model $model" >> $OUT
    PK_MI355_TAIL_LANE=$1 PK_MI355_TAIL_LITE=$2 PK_MI355_CHUNK=$3 timeout -k 10 300 python bench.py --model $model --precision f16x3 --steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s alg  stages %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved'], {k: round(v, 3) for k, v in d['stage_ms_per_step'].items()}))" >> $OUT
  done
done
done
cat $OUT
