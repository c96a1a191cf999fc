# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/soak.log
for i in 1 2 3 4 5; do
(timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"best_sse": [0-9.]*\|"wall_clock_to_optimum_s": [0-9.]*' | tr '\n' ' ' >> gpurun_out/soak.log); echo " run $i rc=$?" >> gpurun_out/soak.log
done
cat gpurun_out/soak.log
