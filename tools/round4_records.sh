#!/bin/bash
# What produced the round-4 records under profiles/ (one MI355X per gpurun call; each part fits one call):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/round4_records.sh PART [TAG]'
# PART = suite | bench | profile | counters | replays | fuzz | calib | exit_cost      (A/Bs of a development-build knob: tools/ab.sh)
# Summaries come back through gpurun_out/ (gpurun_out/profiles/ for the profile parts); copy what is to be judged into profiles/.
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
PART=${1:?part}; TAG=${2:-r04}
mkdir -p gpurun_out
case "$PART" in
  suite)     # the whole GPU suite (shipped build; the dev_knobs tests in one child process on the development build), then smoke()
    bash tools/run_gputests.sh "$TAG" ;;
  bench)     # the line the driver records
    timeout -k 10 900 python3 bench.py > "gpurun_out/${TAG}_bench.json" 2> "gpurun_out/${TAG}_bench.err" || { tail -20 "gpurun_out/${TAG}_bench.err"; exit 1; }
    python3 tools/bench_pick.py < "gpurun_out/${TAG}_bench.json" ;;
  profile)   # bench + rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes per leg -> profiles/bench_pmc.json
    bash tools/gpu_profile.sh "$TAG" 2 ;;
  counters)  # SQ / TCP / TA passes per leg (TA counters one per pass) -> profiles/bench_pmc_extra.json via tools/pmc_extra_summary.py
    bash tools/pmc_extra.sh "$TAG" "1 2 4 3 5 6" headline dragon trimmed ;;
  replays)   # bench.py's N > 1 path rehearsed with 2 ranks on this GPU (gloo), then the N-rank replays (own stderr per run, python's status)
    (timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --rehearse-on-one-gpu > "gpurun_out/${TAG}_rehearse2.log" 2>&1; echo "exit $?" >> "gpurun_out/${TAG}_rehearse2.log")
    tail -1 "gpurun_out/${TAG}_rehearse2.log"
    tools/scale_replays.sh "$TAG" \
      "X=0 8 bunny 5e-5 0.005 2" "X=0 8 dragon 5e-6 0.005 1" "X=0 4 dragon 5e-6 0.005 1" "X=0 2 dragon 5e-6 0.005 1" "X=0 4 bunny 5e-5 0.005 2" "X=0 2 bunny 5e-5 0.005 2" \
      "FGOICP_REPLAY_SCHEDULE=serial 8 bunny 5e-5 0.005 2" "FGOICP_REPLAY_SCHEDULE=serial 8 dragon 5e-6 0.005 1" \
      "FGOICP_REPLAY_TRIM=0.2 8 synthetic1m_outliers 1e-3 0.005 0" ;;
  fuzz)      # randomised differential campaigns against the oracle (operators incl. thresholds; whole runs: SERIAL counters, 2-4 ranks, early exit on / off)
    timeout -k 10 500 python3 tools/fuzz_gpu.py 200 51 > "gpurun_out/${TAG}_fuzz_ops.txt" 2>&1; echo "ops exit $?"; tail -1 "gpurun_out/${TAG}_fuzz_ops.txt"
    timeout -k 10 500 python3 tools/fuzz_gpu.py 60 53 run > "gpurun_out/${TAG}_fuzz_runs.txt" 2>&1; echo "runs exit $?"; tail -1 "gpurun_out/${TAG}_fuzz_runs.txt" ;;
  calib)     # issue rates of the instruction kinds of the bounds kernel; workgroup dispatch rate (the floor under a work item that ends early)
    for b in valu_rate dispatch_rate; do
      [ -x tools/calib/$b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/calib/$b tools/calib/$b.hip || exit 1
      timeout -k 10 120 tools/calib/$b > "gpurun_out/${TAG}_$b.txt" 2>&1; echo "$b exit $?"; tail -12 "gpurun_out/${TAG}_$b.txt"
    done ;;
  exit_cost) # what the early exit's bookkeeping costs on a fixed tick (development build: FGOICP_CUT_PROBE takes it apart)
    export FGOICP_LIB=$PWD/fast-go-icp_amd/lib/libfgoicp_amd_dev.so
    OUT="gpurun_out/${TAG}_early_exit_cost.txt"; : > "$OUT"
    run() { echo "== $*" | tee -a "$OUT"; env "$@" timeout -k 10 300 python3 tools/op_bench.py bunny 256 5 2>> "gpurun_out/${TAG}_early_exit_cost.err" | tee -a "$OUT"; }
    run A=exact; run OP_BENCH_CUT=2; run OP_BENCH_CUT=2 FGOICP_CUT_PROBE=1; run OP_BENCH_CUT=2 FGOICP_CUT_PROBE=2; run OP_BENCH_CUT=2 FGOICP_CUT_PROBE=3
    run OP_BENCH_CUT=0.5; run OP_BENCH_CUT=0.5 FGOICP_CUT_PROBE=4; run OP_BENCH_CUT=0.1; run OP_BENCH_CUT=0.0 ;;
  *) echo "unknown part $PART"; exit 2 ;;
esac
