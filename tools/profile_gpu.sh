#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 passes behind profiles/rNN_*.
#   kernel-trace + stats of the benchmark command in both precision modes, then separate PMC
#   passes (one counter set each, never together with a trace domain other than kernel-trace),
#   then the plain benchmark.  Raw output -> gpurun_out/prof/; tools/make_profiles.py turns it
#   into the small committed summaries.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
B="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-precision --no-other-configs"
B1="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-precision --no-other-configs"
rocprofv3 --kernel-trace --stats -d $OUT/f32 -o p --output-format csv -- python3 $B > $OUT/f32_bench.json 2> $OUT/f32.err
echo "f32 trace done"
rocprofv3 --kernel-trace --stats -d $OUT/f16x3 -o p --output-format csv -- python3 $B --precision f16x3 > $OUT/f16x3_bench.json 2> $OUT/f16x3.err
echo "f16x3 trace done"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- python3 $B1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- python3 $B1 > /dev/null 2> $OUT/pmc_write.err
echo "traffic passes done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d $OUT/pmc_mfma -o p --output-format csv -- python3 $B1 > /dev/null 2> $OUT/pmc_mfma.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma_f16 -o p --output-format csv -- python3 $B1 --precision f16x3 > /dev/null 2> $OUT/pmc_mfma_f16.err
echo "mfma passes done"
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
# keep what travels back small: the per-dispatch traces are not needed, the stats are
find $OUT -name "*kernel_trace.csv" -delete
echo "bench done"
