#!/bin/bash
# ON THE GPU BOX: a sweep of benchmark shapes (sanity: every line must come out; not a measurement of record)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/sweep.txt
: > $OUT
LEAN="--no-cpu-baseline --no-other-configs --no-host-endpoints --steps 3 --warmup 1"
run() {
  echo "## $*" >> $OUT
  timeout -k 10 400 python3 bench.py $LEAN "$@" 2>> gpurun_out/sweep.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d.get('other_precision') or {}
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s  frames %d | other precision %s: %.3f M' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved'], d['config']['frames_per_gpu_per_step'], o.get('precision'), o.get('value', 0)/1e6))" >> $OUT || echo "FAILED" >> $OUT
}
run --batch 64 --seconds 40
run --batch 1024 --seconds 2.5
run --batch 300 --seconds 7.3 --ragged
run --batch 2000 --seconds 0.5
run --batch 7 --seconds 3
run --model W --batch 128 --seconds 12.5
run --softmax reference --batch 128 --no-other-precision
cat $OUT
