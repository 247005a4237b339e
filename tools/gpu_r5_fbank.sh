#!/bin/bash
# round 5 (ON THE GPU BOX): front-end instruction counters of FbankKernel on the headline workload
#   -> gpurun_out/r5_fbank/counters.txt (per launch: duration, SQ_INSTS_VALU / SQ_INSTS_LDS / bank conflicts / CU-busy cycles)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r5_fbank
rm -rf $OUT; mkdir -p $OUT
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/a -o p --output-format csv -- python3 bench.py $COMMON > /dev/null 2> $OUT/a.err || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d $OUT/b -o p --output-format csv -- python3 bench.py $COMMON > /dev/null 2> $OUT/b.err || exit 1
python3 -m pocketkaldi_amd.build --hashes > $OUT/counters.txt
for d in a b; do python3 tools/pmc_summary.py $OUT/$d | grep -A7 "FbankKernel\|CmvnKernel" >> $OUT/counters.txt; done
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/counters.txt
