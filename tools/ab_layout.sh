# Packed-LUT layout of the sparse (bunny-shape) context at the FINAL tick sizes (VERDICT r02 #8a): plain fp32 (203 MB, fits the 256 MiB
# Infinity Cache), z-pair (406 MB), yz-quad (812 MB: the default).  Headline leg: subcubes/s, bounds kernel us/launch, then FETCH_SIZE per launch.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for Z in 2 1 0; do
  FGOICP_LUT_ZPAIR=$Z timeout -k 10 300 python bench.py --only headline > gpurun_out/r3_layout_$Z.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r3_layout_$Z.log') if x.startswith('{"metric"')]
d=json.loads(l[-1]); r=d['roofline']
print('layout=$Z', 'subcubes/s', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'kernel_us', round(r['avg_launch_us'],1), 'launches', r['launches'], 'algorithmic_GBps', round(r['achieved']))
PY
done
cd /tmp
for Z in 2 1 0; do
  rm -rf /tmp/lf_$Z
  FGOICP_LUT_ZPAIR=$Z timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/lf_$Z -- python3 $GRAFT_REPO_ROOT/bench.py --only headline --steps 1 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r3_layout_fetch_$Z.log 2>&1 || { echo "fetch pass $Z failed"; continue; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_generic.py /tmp/lf_$Z.json /tmp/lf_$Z | grep "bounds_sorted_kernel" | sed "s/^/layout=$Z /"
done
