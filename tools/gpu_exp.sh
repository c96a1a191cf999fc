cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t26.log 2>&1; echo "exit $?" >> gpurun_out/t26.log
tail -5 gpurun_out/t26.log
grep -q "exit 0" gpurun_out/t26.log || exit 1
B="python bench.py --no-cpu-baseline --no-dragon --no-trimmed --steps 3 --warmup 1"
run() { echo "== $*" >> gpurun_out/exp26.log; (env "$@" timeout -k 10 200 $B 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"seconds_icp_rank0": [0-9.]*\|"wall_clock_to_optimum_s": [0-9.]*' | tr '\n' ' ' >> gpurun_out/exp26.log); echo >> gpurun_out/exp26.log; }
rm -f gpurun_out/exp26.log
run FGOICP_ICP_OVERLAP=0
run FGOICP_ICP_OVERLAP=1
run FGOICP_ICP_OVERLAP=0
run FGOICP_ICP_OVERLAP=1
for V in 0 1; do
echo "dragon overlap $V" >> gpurun_out/exp26.log
(FGOICP_ICP_OVERLAP=$V timeout -k 10 200 python tools/dragon_probe.py 0 2>&1 | grep -o '"seconds": [0-9.]*\|"seconds_icp": [0-9.]*' | tr '\n' ' ' >> gpurun_out/exp26.log); echo >> gpurun_out/exp26.log
done
cat gpurun_out/exp26.log
