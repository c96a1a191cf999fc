cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03o_gputests.log 2>&1; echo "exit $?" >> gpurun_out/r03o_gputests.log)
tail -3 gpurun_out/r03o_gputests.log
grep -q '^exit 0' gpurun_out/r03o_gputests.log || exit 1
timeout -k 10 900 bash tools/ab_serial_ahead.sh
