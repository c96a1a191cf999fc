cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t11.log 2>&1; echo "pytest exit $?" >> gpurun_out/t11.log
tail -5 gpurun_out/t11.log
rm -f gpurun_out/mb11.log
for z in 1 0; do echo "zpair $z" >> gpurun_out/mb11.log; FGOICP_LUT_ZPAIR=$z timeout -k 10 120 python tests/gpu_microbench.py bunny 0.005 siblings 2>&1 | grep "G= " >> gpurun_out/mb11.log; FGOICP_LUT_ZPAIR=$z timeout -k 10 120 python tests/gpu_microbench.py dragon 0.005 random 2>&1 | grep "G= 64" >> gpurun_out/mb11.log; done
cat gpurun_out/mb11.log
(timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench7.log 2>&1; echo "exit $?" >> gpurun_out/bench7.log)
