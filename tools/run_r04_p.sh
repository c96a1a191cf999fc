#!/bin/bash
# Early exit (fgoicp_bounds_submit_cut): operator test, whole-run tests, then the bench headline (which carries its own full-evaluation comparison).
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
(timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -x -q -k "early_exit or twin or sort or bounds" -s > gpurun_out/r04p_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04p_tests.log)
tail -12 gpurun_out/r04p_tests.log | cut -c1-300
grep -q '^exit 0' gpurun_out/r04p_tests.log || exit 1
timeout -k 10 600 python3 bench.py --only headline > gpurun_out/r04p_bench_headline.json 2> gpurun_out/r04p_bench_headline.err || { tail -20 gpurun_out/r04p_bench_headline.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04p_bench_headline.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", d["value"], d["ms_per_step"], "frac", r["frac"], "launch us", r["avg_launch_us"], "evaluated", r["work_items_evaluated_frac"])
print("full", json.dumps(d.get("full_evaluation"))[:900])
PY
