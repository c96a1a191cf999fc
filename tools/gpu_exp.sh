set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/t2.log 2>&1 || { tail -30 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
for P in 1 2 4 8; do FGOICP_PTS_PER_THREAD=$P timeout -k 10 120 python tests/gpu_microbench.py bunny 0.005 random >> gpurun_out/mb2.log 2>&1; done
FGOICP_PTS_PER_THREAD=4 timeout -k 10 120 python tests/gpu_microbench.py bunny 0.005 siblings >> gpurun_out/mb2.log 2>&1
FGOICP_PTS_PER_THREAD=4 timeout -k 10 120 python tests/gpu_microbench.py bunny 0.02 random >> gpurun_out/mb2.log 2>&1
for P in 4 8; do FGOICP_PTS_PER_THREAD=$P timeout -k 10 300 python tests/gpu_microbench.py dragon 0.005 random --ops >> gpurun_out/mb2.log 2>&1; done
cat gpurun_out/mb2.log
