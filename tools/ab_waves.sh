# A/B (round 3): resident waves per SIMD of the sorted bounds kernel through the register budget (amdgpu_waves_per_eu; default build: 106 VGPRs = 4 waves).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_bounds_waves.txt
: > $OUT
leg() {  # leg name, json key
  python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='' else d['$2']
rf=r.get('roofline') or {}
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, subcubes/s', round(r.get('subcubes_per_s', d['value'])), ', bounds kernel us', round(rf.get('avg_launch_us',0),1), 'best_sse', r.get('best_sse', (d.get('result') or {}).get('best_sse')))"
}
for W in 0 5 6 0 5; do
  unset FGOICP_LIB
  if [ $W != 0 ]; then
    LIB=/tmp/libfgoicp_w$W.so
    [ -f $LIB ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -DFGOICP_BOUNDS_WAVES=$W -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -shared -o $LIB \
       fast-go-icp_amd/csrc/device/kernels.hip fast-go-icp_amd/csrc/device/ctx.hip fast-go-icp_amd/csrc/device/bvh.hip fast-go-icp_amd/csrc/host/solver.cpp fast-go-icp_amd/csrc/host/multi.cpp -ldl 2>/dev/null || exit 1
    export FGOICP_LIB=$LIB
  fi
  echo "== waves per SIMD requested: $W (0 = default build)" | tee -a $OUT
  leg headline "" | tee -a $OUT
  leg dragon dragon_shape | tee -a $OUT
  leg trimmed trimmed_1m_outliers | tee -a $OUT
done
