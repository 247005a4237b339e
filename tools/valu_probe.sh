#!/bin/bash
# Runs ON THE GPU BOX: vector-instruction counters of the non-GEMM kernels (fbank, CMVN, tail) on the
# headline workload -> gpurun_out/prof/S_pmc_valu/.  Own pass, counters only (no trace domain but kernel dispatch).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints"
rm -rf $OUT/S_pmc_valu $OUT/S_pmc_valu2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/S_pmc_valu -o p --output-format csv -- python3 bench.py $COMMON > /dev/null 2> $OUT/S_pmc_valu.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d $OUT/S_pmc_valu2 -o p --output-format csv -- python3 bench.py $COMMON > /dev/null 2> $OUT/S_pmc_valu2.err
echo valu_probe done
