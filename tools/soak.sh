#!/bin/bash
# Soak run of the seeded fuzz tests on the GPU box: many more seeds than the default suite, in one
# pytest process.  usage: tools/soak.sh [count=300] [base=1000] [light|heavy]   (writes gpurun_out/soak.log)
#   light (default): layer stacks, decodable contexts, ragged batches, f16x3 -- ~50 cases per second
#   heavy: the big-tile decodable and wide-model batch tests -- ~2 cases per second
mkdir -p gpurun_out
SEL='fuzz and not fuzz_big and not fuzz_batches'
[ "${3:-light}" = heavy ] && SEL='fuzz_big or fuzz_batches'
PK_FUZZ_SEEDS=${1:-300} PK_FUZZ_BASE=${2:-1000} python -m pytest tests/test_gpu_parity.py tests/test_gpu_f16_range.py -m gpu -q -k "$SEL" \
  -p no:cacheprovider > gpurun_out/soak.log 2>&1
rc=$?
tail -15 gpurun_out/soak.log
exit $rc
