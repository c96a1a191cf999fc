cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_trimming.py -m gpu -x -q 2>&1 | tail -2
for V in 1 2; do (FGOICP_TRIM_VARIANT=$V timeout -k 10 300 python tools/run_probe.py 0 1e-3 synthetic1m_outliers 0.2 150 2>&1 | grep -o '"seconds": [0-9.]*\|"kernel_ms": [0-9.]*\|"seconds_icp": [0-9.]*\|"best_sse": [0-9.]*' | tr '\n' ' '); echo " trim_variant=$V"; done
