#!/bin/bash
# fbank mel stage: bit-exact tests + stage time old tree / new tree (model tiny keeps the run short; fbank does not depend on the model)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_refmodel_files.py -m gpu -x -q -k "fbank or fft or logf or silence or non_integer or lengths or whole_path or full_size_batch_properties or refmodel or int16" > gpurun_out/r4f_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4f_tests.log
tail -5 gpurun_out/r4f_tests.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4f_fbank_ab.txt
: > $OUT
for rep in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then cd $GRAFT_REPO_ROOT/gpurun_ab/old_tree; else cd $GRAFT_REPO_ROOT; fi
  echo "## $v" >> $OUT
  timeout -k 10 300 python bench.py --model tiny --steps 10 --warmup 3 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.3f  stages %s' % (d['ms_per_step'], {k: round(v, 4) for k, v in d['stage_ms_per_step'].items()}))" >> $OUT
done
done
cat $OUT
