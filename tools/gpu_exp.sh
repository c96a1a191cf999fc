# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "TCC_EA0_RD|TCC_EA0_WR|TCC_BUBBLE|TCC_REQ|TCC_READ|FETCH_SIZE|TCC_MC_RD" | sort -u | head -60 > gpurun_out/r02e_counters.txt
cat gpurun_out/r02e_counters.txt | cut -c1-220
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/calib/fetch_calib.hip 2>/dev/null
cd /tmp && rm -rf /tmp/calib2 && timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d /tmp/calib2 -- /tmp/fetch_calib > $REPO/gpurun_out/r02e_calib.log 2>&1
tail -3 $REPO/gpurun_out/r02e_calib.log
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/calib2/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(rows.items()):
    print(k, {c: x[:2] for c, x in v.items()})
PY
