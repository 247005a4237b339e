#!/bin/bash
# round 5 (ON THE GPU BOX): the ragged last column tile of the fused-tail launch as a 64-column strip launch -- parity, then A/B.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r5_strip.txt
: > $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused_tail or nan_rows or wide_model or whole_path" > gpurun_out/r5_strip_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r5_strip_tests.log
[ $rc -eq 0 ] || exit $rc
LEAN="--no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints"
for rep in 1 2 3; do for v in 0 1; do
  echo "## PK_MI355_TAIL_STRIP=$v" >> $OUT
  PK_MI355_TAIL_STRIP=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 $LEAN 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['stage_ms_per_step']['gemm']))" >> $OUT || exit 1
done; done
for v in 0 1; do
  echo "## model W, PK_MI355_TAIL_STRIP=$v" >> $OUT
  PK_MI355_TAIL_STRIP=$v timeout -k 10 300 python3 bench.py --model W --steps 5 --warmup 2 $LEAN 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['stage_ms_per_step']['gemm']))" >> $OUT || exit 1
done
cat $OUT
