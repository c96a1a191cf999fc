# The randomised differential campaign (tools/fuzz_gpu.py) under the k-d orders of round 3 (defaults) and their A/B settings.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_fuzz_kd.txt
: > $OUT
run() { echo "== $*" | tee -a $OUT; env "$@" python tools/fuzz_gpu.py 300 41 2>&1 | tail -n 2 | tee -a $OUT; }
echo "== defaults (k-d target tree, k-d source order, in-run order), 600 operator cases, seed 9" | tee -a $OUT
python tools/fuzz_gpu.py 600 9 2>&1 | tail -n 2 | tee -a $OUT
echo "== defaults, 60 whole runs, seed 32" | tee -a $OUT
python tools/fuzz_gpu.py 60 32 run 2>&1 | tail -n 2 | tee -a $OUT
run FGOICP_BVH_ORDER=0 FGOICP_POINT_CURVE=1
run FGOICP_POINT_CURVE=3
run FGOICP_KD_FINE=0 FGOICP_ICP_DUAL=1
echo "== FUZZ_SCALE=16 (40 cases)" | tee -a $OUT; FUZZ_SCALE=16 python tools/fuzz_gpu.py 40 23 2>&1 | tail -n 2 | tee -a $OUT
