#!/bin/bash
# A/B of the working tree against a full copy of an older tree (gpurun_ab/old_tree, its own library inside) on one box.
# usage: tools/ab_tree.sh "<bench args>"   -> gpurun_out/ab_tree.txt
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_tree.txt
: > $OUT
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then cd $GRAFT_REPO_ROOT/gpurun_ab/old_tree; else cd $GRAFT_REPO_ROOT; fi
  for model in W S; do
    echo "## $v model $model" >> $OUT
    timeout -k 10 300 python bench.py --model $model $1 --steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s alg  stages %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved'], {k: round(v, 3) for k, v in d['stage_ms_per_step'].items()}))" >> $OUT
  done
done
done
cat $OUT
