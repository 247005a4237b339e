#!/bin/bash
# round 4: does a second lane (two chunks' layer stacks on two streams) still pay now that the tail is fused?
set -e
OUT=gpurun_out/r4l_lanes.txt
: > $OUT
PREC=f32
COMMON="--steps 10 --warmup 3 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints"
run() {  # $1 = label, rest = env
  echo "## $1" >> $OUT
  for i in 1 2; do
    env "${@:2}" python3 bench.py $COMMON --precision $PREC 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f' % (d['value']/1e6, d['ms_per_step']))" >> $OUT
  done
}
run "default (one lane, chunk 262144)" PK_X=0
run "LANES=2 CHUNK=131072" PK_MI355_LANES=2 PK_MI355_CHUNK=131072
run "LANES=2 CHUNK=65536" PK_MI355_LANES=2 PK_MI355_CHUNK=65536
run "LANES=2 CHUNK=32768" PK_MI355_LANES=2 PK_MI355_CHUNK=32768
run "LANES=1 CHUNK=131072" PK_MI355_CHUNK=131072
PREC=f16x3
run "f16x3 default" PK_X=0
run "f16x3 LANES=2 CHUNK=131072" PK_MI355_LANES=2 PK_MI355_CHUNK=131072
cat $OUT
