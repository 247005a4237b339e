#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_exact_tail.txt
: > $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_refmodel_files.py tests/test_gpu_decoder.py -m gpu -x -q -k "reference_softmax or refmodel or reference_written or real_decoder or whole_path" >> $OUT 2>&1 || { tail -30 $OUT; exit 1; }
for rep in 1 2; do
for two in "" 1; do
echo "## PK_MI355_EXACT_TAIL_TWO_PASS=$two" >> $OUT
PK_MI355_EXACT_TAIL_TWO_PASS=$two timeout -k 10 300 python bench.py --softmax reference --steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('reference softmax: value %.3f M frames/s  ms/step %.3f  stages %s' % (d['value']/1e6, d['ms_per_step'], d['stage_ms_per_step']))" >> $OUT
done
done
cat $OUT
