#!/bin/bash
# The whole GPU suite (shipped build in this process, the dev_knobs tests in one child process on the development build), then smoke.
#   tools/run_gputests.sh TAG
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04}
mkdir -p gpurun_out
(timeout -k 10 1700 python3 -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputests.log 2>&1; echo "exit $?" >> gpurun_out/${TAG}_gputests.log)
tail -6 gpurun_out/${TAG}_gputests.log | cut -c1-400
grep -q '^exit 0' gpurun_out/${TAG}_gputests.log || exit 1
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
