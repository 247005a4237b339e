set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(rocprofv3 --list-avail > gpurun_out/list_avail.txt 2>&1 || true)
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r3a.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest_r3a.log
tail -30 gpurun_out/gputest_r3a.log
