#!/bin/bash
# what does the placement of the activation operands cost in clock?  one box, alternating:
# uncalibrated (exponents 0), calibrated to [2^3, 2^4) (shipped), calibrated to [2^10, 2^11) (the first choice)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_f16_range.py -m gpu -x -q 2>&1 | tail -2
OUT=gpurun_out/r4i_calibration_ab.txt
: > $OUT
for rep in 1 2 3; do
for cfg in "1 3" "0 3" "0 10"; do
  set -- $cfg
  for model in W S; do
    echo "## PK_BENCH_NO_CALIBRATE=$1 PK_MI355_CALIB_TARGET_LOG2=$2 model $model" >> $OUT
    PK_BENCH_NO_CALIBRATE=$1 PK_MI355_CALIB_TARGET_LOG2=$2 timeout -k 10 300 python bench.py --model $model --precision f16x3 --steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s alg  exps %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved'], d['config']['operand_exponents_calibrated_on_root']))" >> $OUT
  done
done
done
cat $OUT
