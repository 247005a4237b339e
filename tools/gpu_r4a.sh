#!/bin/bash
# round-4 step 1: range-safety tests + f16x3 parity + throughput A/B old/new library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_f16_range.py tests/test_gpu_parity.py -m gpu -x -q -k "f16 or range or calib or saturat or weight_scale or exponent or broadcast or wide or fuzz" > gpurun_out/r4a_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4a_tests.log
tail -25 gpurun_out/r4a_tests.log
bash tools/ab_tree.sh "--precision f16x3" > gpurun_out/r4a_ab.log 2>&1
tail -12 gpurun_out/r4a_ab.log
