#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f16_range.py tests/test_refmodel_files.py -m gpu -x -q -k "f16 or wide or full_size or fuzz_f16 or refmodel or range or wider or calib or rescaled" > gpurun_out/r4h_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4h_tests.log
tail -6 gpurun_out/r4h_tests.log
grep -q "rc=0" gpurun_out/r4h_tests.log || exit 1
bash tools/ab_tree.sh "--precision f16x3" > gpurun_out/r4h_ab.log 2>&1
cat gpurun_out/ab_tree.txt
