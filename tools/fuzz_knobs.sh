# The randomised differential campaign (tools/fuzz_gpu.py) under the round's knobs: every new kernel / loop variant against the oracle on ragged inputs.
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python tools/fuzz_gpu.py 250 21 2>&1 | tail -n 2; }
python tools/fuzz_gpu.py 600 8 2>&1 | tail -n 1
python tools/fuzz_gpu.py 60 31 run 2>&1 | tail -n 2
run FGOICP_UNITS=4
run FGOICP_UNITS=8 FGOICP_ICP_DUAL=1
run FGOICP_LDS_TILES=128
run FGOICP_NN_FLAT=1 FGOICP_NN_CLAIM=0
run FGOICP_ICP_DEVICE=1
run FGOICP_ICP_FUSE=0 FGOICP_ICP_DUAL=0
echo "== FUZZ_SCALE=16 FGOICP_ICP_DUAL=1 (30 cases)"; FUZZ_SCALE=16 FGOICP_ICP_DUAL=1 python tools/fuzz_gpu.py 30 22 2>&1 | tail -n 2
