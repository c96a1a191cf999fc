cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tests/gpu_nn_timing.py bunny > gpurun_out/nn1.log 2>&1; cat gpurun_out/nn1.log
timeout -k 10 300 python tests/gpu_nn_timing.py dragon > gpurun_out/nn2.log 2>&1; cat gpurun_out/nn2.log
