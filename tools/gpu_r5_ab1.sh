#!/bin/bash
# round 5, experiment record 1 (ON THE GPU BOX): the fused-tail launch's memory traffic and the first layer's alignment.
#   variants (gpurun_ab/*.so, built by tools/build_variant.py from the same tree):
#     base   the product
#     nt     -DPK_EXP_TAIL_NT: logits stores `sc1 nt`, the tail's loads `sc1 nt` (streaming hints: keep the operand panels in L2)
#     diag   -DPK_MI355_DIAG with PK_DEBUG_TAILFLAGS=16: the hand-off without the tail phases (NO results) -- what the phases cost in traffic
#     align  -DPK_EXP_SPLICE_ALIGNED: TIMING ONLY, every splice shift rounded down to a multiple of four columns (16-byte aligned DMA rows)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r5_ab1
rm -rf $OUT; mkdir -p $OUT
LEAN="--no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints"
cp pocketkaldi_amd/libpk_mi355.so /tmp/keep.so
trap 'cp /tmp/keep.so pocketkaldi_amd/libpk_mi355.so' EXIT
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "nan_rows" > $OUT/nan_test.log 2>&1; echo "nan test rc=$?" >> $OUT/nan_test.log
grep -E "passed|failed|NaN placement" $OUT/nan_test.log | cut -c1-400
use() { cp gpurun_ab/$1.so pocketkaldi_amd/libpk_mi355.so; }
for rep in 1 2; do
  for v in base nt align; do
    use $v
    echo "## $v" >> $OUT/steps.txt
    timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 $LEAN 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s  stages %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved'], {k: round(v, 3) for k, v in d['stage_ms_per_step'].items()}))" >> $OUT/steps.txt || exit 1
  done
  echo "## walk8 (base, PK_MI355_TAIL_WALK=8)" >> $OUT/steps.txt
  use base
  PK_MI355_TAIL_WALK=8 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 $LEAN 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.3f M frames/s  ms/step %.3f  gemm %.1f TFLOP/s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['achieved']))" >> $OUT/steps.txt || exit 1
done
cat $OUT/steps.txt
# per-kernel durations: base vs align (the first layer is the <..., true (SPLICE), ...> instantiation)
for v in base align; do
  use $v
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$v -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 $LEAN > /dev/null 2> $OUT/trace_$v.err || exit 1
  find $OUT/trace_$v -name "*kernel_trace.csv" -delete
done
# traffic: one counter per pass
B1="bench.py --steps 1 --warmup 1 $LEAN"
for v in base nt; do
  use $v
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $OUT/pmc_${v}_$c -o p --output-format csv -- python3 $B1 > /dev/null 2> $OUT/pmc_${v}_$c.err || exit 1
  done
done
use diag
export PK_DEBUG_TAILFLAGS=16
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc_diag16_$c -o p --output-format csv -- python3 $B1 > /dev/null 2> $OUT/pmc_diag16_$c.err || exit 1
done
unset PK_DEBUG_TAILFLAGS
use base
export PK_MI355_TAIL_WALK=8
for c in FETCH_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc_walk8_$c -o p --output-format csv -- python3 $B1 > /dev/null 2> $OUT/pmc_walk8_$c.err || exit 1
done
unset PK_MI355_TAIL_WALK
for d in $OUT/pmc_*; do [ -d $d ] && { echo "== $d"; python3 tools/pmc_summary.py $d | grep -A3 "GemmKernel"; } >> $OUT/pmc.txt; done
find $OUT -name "*counter_collection.csv" -size +2M -delete
cat $OUT/pmc.txt | head -120
echo "r5_ab1 done"
