#!/bin/bash
# ON THE GPU BOX: A/B of one environment switch on the headline workload, alternating, three pairs.
#   usage: tools/gpu_ab_env.sh VAR value_a value_b [extra bench args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
VAR=$1; A=$2; B=$3; shift 3
OUT=gpurun_out/ab_env_$VAR.txt
: > $OUT
LEAN="--no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints"
for rep in 1 2 3; do for v in $A $B; do
  echo "## $VAR=$v" >> $OUT
  env $VAR=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 $LEAN "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  stages %s' % (d['value']/1e6, d['ms_per_step'], {k: round(v, 3) for k, v in d['stage_ms_per_step'].items()}))" >> $OUT || exit 1
done; done
cat $OUT
