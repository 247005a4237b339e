#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_tail_wave.txt
: > $OUT
PK_MI355_TAIL_WAVE=8 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_path or batch_ragged or decodable_tiny or softmax_prob or full_size_batch_properties" >> $OUT 2>&1 || { tail -20 $OUT; exit 1; }
for rep in 1 2; do
for w in 0 4 8 16 32; do
  echo "## PK_MI355_TAIL_WAVE=$w" >> $OUT
  PK_MI355_TAIL_WAVE=$w timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  stages %s' % (d['value']/1e6, d['ms_per_step'], {k: round(v, 3) for k, v in d['stage_ms_per_step'].items()}))" >> $OUT
done
done
cat $OUT
