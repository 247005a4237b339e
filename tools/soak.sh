#!/bin/bash
# Soak run of the seeded fuzz tests on the GPU box: many more seeds than the default suite, in one
# pytest process.  usage: tools/soak.sh [count=300] [base=1000]    (writes gpurun_out/soak.log)
mkdir -p gpurun_out
PK_FUZZ_SEEDS=${1:-300} PK_FUZZ_BASE=${2:-1000} python -m pytest tests/test_gpu_parity.py -m gpu -q -k fuzz \
  -p no:cacheprovider > gpurun_out/soak.log 2>&1
rc=$?
tail -15 gpurun_out/soak.log
exit $rc
