#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 passes behind profiles/rNN_*.
#   usage: tools/profile_gpu.sh [S] [W] [B1] [bench]      (default: all four sections)
#   S     model S, 256 utt x 10 s (the headline workload): kernel-trace + stats in both precision
#         modes, then separate PMC passes (one counter set each, never together with a trace
#         domain other than kernel-trace): FETCH_SIZE, WRITE_SIZE, MFMA busy
#   W     BASELINE configs[4], the wide model 440 -> 6 x 2048 -> 8000: the same passes
#   B1    BASELINE configs[1], one utterance: kernel-trace + stats of tools/b1_probe.py
#   bench the plain benchmark line
# Raw output -> gpurun_out/prof/; tools/make_profiles.py turns it into the small committed summaries.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
SECTIONS="${*:-S W B1 bench}"
COMMON="--no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints"
PMC_MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"

passes() {   # $1 = tag (S / W), $2 = extra bench args
  local tag=$1 extra=$2
  local B="bench.py --steps 5 --warmup 2 $COMMON $extra"
  local B1="bench.py --steps 1 --warmup 1 $COMMON $extra"
  for mode in f32 f16x3; do
    rm -rf $OUT/${tag}_$mode
    rocprofv3 --kernel-trace --stats -d $OUT/${tag}_$mode -o p --output-format csv -- python3 $B --precision $mode > $OUT/${tag}_${mode}_bench.json 2> $OUT/${tag}_$mode.err
    echo "$tag $mode trace done"
    rm -rf $OUT/${tag}_pmc_fetch_$mode $OUT/${tag}_pmc_write_$mode $OUT/${tag}_pmc_mfma_$mode
    rocprofv3 --pmc FETCH_SIZE -d $OUT/${tag}_pmc_fetch_$mode -o p --output-format csv -- python3 $B1 --precision $mode > /dev/null 2> $OUT/${tag}_pmc_fetch_$mode.err
    rocprofv3 --pmc WRITE_SIZE -d $OUT/${tag}_pmc_write_$mode -o p --output-format csv -- python3 $B1 --precision $mode > /dev/null 2> $OUT/${tag}_pmc_write_$mode.err
    rocprofv3 --pmc $PMC_MFMA -d $OUT/${tag}_pmc_mfma_$mode -o p --output-format csv -- python3 $B1 --precision $mode > /dev/null 2> $OUT/${tag}_pmc_mfma_$mode.err
    echo "$tag $mode PMC passes done"
  done
}

for s in $SECTIONS; do
  # what this section measures: hashes of the library's sources (no .git on the GPU box)
  python3 -m pocketkaldi_amd.build --hashes > $OUT/$s.head
  case $s in
    S) passes S "" ;;
    W) passes W "--model W" ;;
    B1) rm -rf $OUT/B1
        rocprofv3 --kernel-trace --stats -d $OUT/B1 -o p --output-format csv -- python3 tools/b1_probe.py > $OUT/B1_probe.log 2> $OUT/B1.err
        echo "B1 trace done" ;;
    bench) python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench done" ;;
  esac
done
# keep what travels back small: the per-dispatch traces are not needed, the stats are
find $OUT -name "*kernel_trace.csv" -delete
echo "profile_gpu done: $SECTIONS"
