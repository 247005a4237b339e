#!/bin/bash
# f16x3 tile-walk A/B on the wide model: step time + per-kernel FETCH_SIZE (separate PMC pass), per walk shape
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4e_walk_ab.txt
: > $OUT
COMMON="--model W --precision f16x3 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints"
for walk in 4x4 8x4 4x8 2x16 16x2 8x8 1x32 2x8; do
  echo "## PK_MI355_F16_WALK=$walk" >> $OUT
  for rep in 1 2; do
  PK_MI355_F16_WALK=$walk timeout -k 10 300 python bench.py $COMMON --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s alg  gemm ms %.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved'], d['stage_ms_per_step']['gemm']))" >> $OUT
  done
  rm -rf gpurun_out/prof_walk
  PK_MI355_F16_WALK=$walk timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_walk -o p --output-format csv -- python3 bench.py $COMMON --steps 1 --warmup 1 > /dev/null 2> gpurun_out/prof_walk.err
  python tools/pmc_summary.py gpurun_out/prof_walk | grep -A1 "GemmF16K32" >> $OUT
done
rm -rf gpurun_out/prof_walk
cat $OUT
