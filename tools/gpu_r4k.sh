#!/bin/bash
# fused fp32 tail as the default: the whole GPU suite, then the headline A/B against the tree before it (gpurun_ab/old_tree)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4k_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4k_tests.log
tail -6 gpurun_out/r4k_tests.log
grep -q "rc=0" gpurun_out/r4k_tests.log || exit 1
bash tools/ab_tree.sh "" > gpurun_out/r4k_ab.log 2>&1
cat gpurun_out/ab_tree.txt
