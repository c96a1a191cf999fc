# A/B (round 3): points per leaf of the target tree (development builds with -DFGOICP_BVH_LEAF=N; default 32).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_bvh_leaf.txt
: > $OUT
leg() {
  python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='' else d['$2']
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, icp ms', round(r['seconds_icp_rank0']*1e3,2), 'setup s', round(r['setup_s_upload_plus_lut_build'],3), 'best_sse', r.get('best_sse', (d.get('result') or {}).get('best_sse')))"
}
for N in 32 16 64 32 16; do
  unset FGOICP_LIB
  if [ $N != 32 ]; then
    LIB=/tmp/libfgoicp_leaf$N.so
    [ -f $LIB ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -DFGOICP_BVH_LEAF=$N -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -shared -o $LIB \
       fast-go-icp_amd/csrc/device/kernels.hip fast-go-icp_amd/csrc/device/ctx.hip fast-go-icp_amd/csrc/device/bvh.hip fast-go-icp_amd/csrc/host/solver.cpp fast-go-icp_amd/csrc/host/multi.cpp -ldl 2>/dev/null || exit 1
    export FGOICP_LIB=$LIB
  fi
  echo "== points per leaf: $N" | tee -a $OUT
  for W in bunny dragon; do
    ICP_VARIANT=default python tools/icp_bench.py $W 5 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  icp_bench', d['workload'], 'thr', d['thr'], 'iters', d['iters'], 'us/iter', round(d['us_per_iter'],1), 'sse', d['sse'])" | tee -a $OUT
  done
  leg default_threshold reference_default_threshold | tee -a $OUT
  leg dragon dragon_shape | tee -a $OUT
  leg trimmed trimmed_1m_outliers | tee -a $OUT
done
