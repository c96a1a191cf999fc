# Timing-only ablation builds of the bounds kernel (cdna_hip_programming.md §7 "Ablate"): what a kernel that hoisted the rotation
# out of the per-evaluation work (TODO.md:11 of the reference, "rotate once") or served every gather from on-chip storage
# (LDS-staged tiles) could gain AT MOST.  Results are wrong by construction; only the kernel times matter.
#   bash tools/ablate.sh   (on the GPU box; writes gpurun_out/profiles/r02_ablation_fixed_tick.txt)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
OUT=gpurun_out/profiles/r02_ablation_fixed_tick.txt
: > $OUT
for AB in 0 1 4 5; do
  LIB=/tmp/libfgoicp_ab$AB.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -DFGOICP_ABLATE=$AB -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -shared -o $LIB \
     fast-go-icp_amd/csrc/device/kernels.hip fast-go-icp_amd/csrc/device/ctx.hip fast-go-icp_amd/csrc/device/bvh.hip fast-go-icp_amd/csrc/host/solver.cpp fast-go-icp_amd/csrc/host/multi.cpp -ldl 2>/dev/null || exit 1
  for WL in bunny dragon; do
    echo "ablate=$AB $(FGOICP_LIB=$LIB timeout -k 10 300 python tools/op_bench.py $WL 1024 5 2>/dev/null)" | tee -a $OUT
  done
done
