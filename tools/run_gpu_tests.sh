set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_r4.log
tail -30 gpurun_out/gputest_r4.log
