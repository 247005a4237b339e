#!/bin/bash
# A/B of two prebuilt libraries (gpurun_ab/old.so, gpurun_ab/new.so) on one box: the library file is swapped under a
# stamp that matches the tree, so nothing is rebuilt.  usage: tools/ab_so.sh "<bench args>" [label]
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_so.txt
: > $OUT
cp pocketkaldi_amd/libpk_mi355.so /tmp/keep.so
trap 'cp /tmp/keep.so pocketkaldi_amd/libpk_mi355.so' EXIT      # an interrupted A/B must not leave a swapped library under the tree's stamp
for rep in 1 2; do
for v in new old; do
  cp gpurun_ab/$v.so pocketkaldi_amd/libpk_mi355.so
  for model in W S; do
    echo "## $v model $model" >> $OUT
    timeout -k 10 300 python bench.py --model $model $1 --steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s alg  stages %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved'], {k: round(v, 3) for k, v in d['stage_ms_per_step'].items()}))" >> $OUT
  done
done
done
cp /tmp/keep.so pocketkaldi_amd/libpk_mi355.so
cat $OUT
