#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the recurring GPU jobs of the build loop, one entry each
# (replaces the per-experiment gpu_r4?.sh scripts).  Output -> gpurun_out/<name>.log / .txt.
#   tools/gpu.sh tests [pytest args]     the -m gpu suite (one process)
#   tools/gpu.sh bench [bench args]      the default benchmark line -> gpurun_out/bench.json
#   tools/gpu.sh quick [bench args]      the headline workload only (no supplementary legs), 3 runs
#   tools/gpu.sh ab "<bench args>" v1.so v2.so ...   A/B of prebuilt libraries under gpurun_ab/ (tools/ab_so.sh's loop,
#                                        any number of variants, headline workload), 2 rounds
#   tools/gpu.sh stress [S passes] [W passes]        tools/fused_tail_stress.py
# Steps of one call are joined with && by the caller; a step that times out ends the call.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
LEAN="--no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints"
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s  stages %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved'], {k: round(v, 3) for k, v in d['stage_ms_per_step'].items()}))"; }
case $1 in
  tests) shift
    timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q "$@" > gpurun_out/gputest.log 2>&1; rc=$?
    echo "pytest rc=$rc" >> gpurun_out/gputest.log; tail -15 gpurun_out/gputest.log; exit $rc ;;
  bench) shift
    timeout -k 10 900 python3 bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
    tail -c 3000 gpurun_out/bench.json; exit $rc ;;
  quick) shift
    : > gpurun_out/quick.txt
    for i in 1 2 3; do timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 $LEAN "$@" 2>/dev/null | line >> gpurun_out/quick.txt || exit 1; done
    cat gpurun_out/quick.txt ;;
  ab) shift; args=$1; shift
    OUT=gpurun_out/ab.txt; : > $OUT
    cp pocketkaldi_amd/libpk_mi355.so /tmp/keep.so
    trap 'cp /tmp/keep.so pocketkaldi_amd/libpk_mi355.so' EXIT     # an interrupted A/B must not leave a swapped library under the tree's stamp
    for rep in 1 2; do for v in "$@"; do
      cp gpurun_ab/$v pocketkaldi_amd/libpk_mi355.so
      echo "## $v" >> $OUT
      timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 $LEAN $args 2>/dev/null | line >> $OUT || { echo "FAILED" >> $OUT; }
    done; done
    cat $OUT ;;
  stress) shift
    timeout -k 10 1100 python3 tools/fused_tail_stress.py "${1:-300}" "${2:-60}" > gpurun_out/stress.log 2>&1; rc=$?
    tail -5 gpurun_out/stress.log; exit $rc ;;
  *) echo "usage: tools/gpu.sh tests|bench|quick|ab|stress ..."; exit 2 ;;
esac
