cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_fuzz_serial_sharded.txt
echo "== whole runs, seed 33, 100 cases: oracle vs SERIAL vs SERIAL on 2-4 ranks (counters and bits) vs ROUND" | tee $OUT
timeout -k 10 900 python tools/fuzz_gpu.py 100 33 run 2>&1 | tail -n 4 | tee -a $OUT
echo "== the same with cooperative refinements split over the ranks (FGOICP_COOP_ICP=1 FGOICP_COOP_SPLIT_MIN=0), seed 34, 60 cases" | tee -a $OUT
FGOICP_COOP_ICP=1 FGOICP_COOP_SPLIT_MIN=0 timeout -k 10 600 python tools/fuzz_gpu.py 60 34 run 2>&1 | tail -n 4 | tee -a $OUT
