#!/bin/bash
# Developer helper: compile one .hip for gfx950, print resource usage and the filtered ISA of a kernel.
# usage: tools/isa.sh gemm.hip 'GemmKernelILb0ELb0' [from] [to]
set -e
SRC=/root/repo/pocketkaldi_amd/csrc/$1
OUT=/tmp/pk_isa; mkdir -p $OUT; cd $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -c $SRC -o $OUT/x.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|Occupancy|LDS Size|Spill" | grep -A5 "$2" | head -8
S=$(ls $OUT/*-hip-amdgcn-amd-amdhsa-gfx950.s | head -1)
awk "/^_ZN.*$2.*:/,/s_endpgm/" $S | grep -E "v_mfma|ds_read|ds_write|s_waitcnt|s_barrier|global_load|global_store|buffer_|s_cbranch|s_setprio" | sed -n "${3:-1},${4:-80}p"
