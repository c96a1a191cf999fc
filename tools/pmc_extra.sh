# What binds the bounds kernel on a leg: SQ / TA / TCP counter passes of `bench.py --only LEG` (separate rocprofv3 --pmc passes,
# --kernel-trace only), summed per kernel by tools/pmc_generic.py.   bash tools/pmc_extra.sh <tag> <passes: e.g. "1 2 3 4"> <leg> [<leg> ...]
cd $GRAFT_REPO_ROOT
REPO=$GRAFT_REPO_ROOT
TAG=$1; shift
PASSES=$1; shift
mkdir -p $REPO/gpurun_out/profiles
export TMPDIR=/tmp
cd /tmp
SET1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
SET2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
# the texture-addresser counters ONE PER PASS: the three together hung this image's rocprofv3 until its time limit in round 3 (twice, every leg);
# one at a time every pass completes (round 4, profiles/r04_dragon_TA_*.json)
SET3="TA_BUSY_avr GRBM_GUI_ACTIVE"
SET5="TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
SET6="TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
SET4="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"
for LEG in "$@"; do
  DIRS=""
  for i in $PASSES; do
    eval SET=\$SET$i
    D=/tmp/px_${LEG}_$i
    rm -rf $D
    timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $D -- python3 $REPO/bench.py --only $LEG --steps 1 --warmup 0 > $REPO/gpurun_out/${TAG}_px_${LEG}_$i.log 2>&1
    rc=$?
    echo "exit status $rc" >> $REPO/gpurun_out/${TAG}_px_${LEG}_$i.log
    # a pass that fails or is killed at its limit ends the whole script (no further GPU step after a kill); its log stays in gpurun_out/
    if [ $rc -ne 0 ]; then echo "pass $i of $LEG ended with status $rc"; tail -n 3 $REPO/gpurun_out/${TAG}_px_${LEG}_$i.log | cut -c1-300; exit 1; fi
    DIRS="$DIRS $D"
    echo "leg $LEG pass $i done"
  done
  python3 $REPO/tools/pmc_generic.py $REPO/gpurun_out/profiles/${TAG}_${LEG}_pmc_extra_p$(echo $PASSES | tr -d ' ').json $DIRS > $REPO/gpurun_out/${TAG}_px_${LEG}_summary_p$(echo $PASSES | tr -d ' ').txt 2>&1
  grep -E "bounds_(item|sorted)_kernel" $REPO/gpurun_out/${TAG}_px_${LEG}_summary_p$(echo $PASSES | tr -d ' ').txt | head -20
done
