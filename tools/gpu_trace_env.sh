#!/bin/bash
# ON THE GPU BOX: per-kernel durations (rocprofv3 --kernel-trace --stats) of the headline workload under two values of one
# environment switch.   usage: tools/gpu_trace_env.sh VAR value_a value_b
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
VAR=$1
OUT=gpurun_out/trace_env_$VAR
rm -rf $OUT; mkdir -p $OUT
LEAN="--no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints"
for v in $2 $3 $2 $3; do
  export $VAR=$v
  rocprofv3 --kernel-trace --stats -d $OUT/t_$v -o p --output-format csv -- python3 bench.py --steps 8 --warmup 2 $LEAN > /dev/null 2> $OUT/t_$v.err || exit 1
  echo "## $VAR=$v" >> $OUT/summary.txt
  python3 - "$(find $OUT/t_$v -name '*kernel_stats.csv' | head -1)" >> $OUT/summary.txt <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(r"GemmKernel|Fbank|Cmvn", r["Name"]):
        name = re.sub(r"^void |pkmi::\(anonymous namespace\)::|\(.*$", "", r["Name"])
        print("  %-60s calls %3s  avg %9.1f us" % (name, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/t_$v
done
cat $OUT/summary.txt
