#!/bin/bash
# fp32 fused tail (PK_MI355_FUSED_TAIL32): parity, then A/B on the headline workload
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fp32_fused" > gpurun_out/r4j_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4j_tests.log
tail -12 gpurun_out/r4j_tests.log
grep -q "rc=0" gpurun_out/r4j_tests.log || exit 1
OUT=gpurun_out/r4j_fused_tail32.txt
: > $OUT
for rep in 1 2 3; do
for cfg in "0 0 384" "1 0 384" "1 0 100000000"; do
  set -- $cfg
  for model in S W; do
    echo "## PK_MI355_FUSED_TAIL32=$1 dbg=$2 min_tiles=$3 model $model" >> $OUT
    PK_DEBUG_TAILTIME=1 PK_DEBUG_TAILFLAGS=$2 PK_MI355_FUSED_TAIL32=$1 PK_MI355_FUSED_TAIL_MIN_TILES=$3 timeout -k 10 300 python bench.py --model $model --steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints 2>gpurun_out/r4j_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm ms %.3f  tail %.3f' % (d['value']/1e6, d['ms_per_step'], d['stage_ms_per_step']['gemm'], d['stage_ms_per_step']['tail']))" >> $OUT
    grep "fused tail phases" gpurun_out/r4j_err.txt | tail -1 >> $OUT
  done
done
done
cat $OUT
