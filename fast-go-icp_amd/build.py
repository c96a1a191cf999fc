"""Builds libfgoicp_amd.so (HIP kernels for gfx950 + C ABI + host driver) in-tree with hipcc.

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the CPU-only container
as the "does it build" check and the resulting .so travels to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
DEFAULT_LIB = os.path.join(LIB_DIR, "libfgoicp_amd.so")
DEV_LIB = os.path.join(LIB_DIR, "libfgoicp_amd_dev.so")  # the same sources with -DFGOICP_DEV_KNOBS (csrc/host/knobs.hpp): reads the A/B environment knobs, instantiates the rejected kernel variants
LIB_PATH = os.environ.get("FGOICP_LIB") or DEFAULT_LIB  # FGOICP_LIB: another build to load (the development build, tools/ablate.sh)
CLI_PATH = os.path.join(LIB_DIR, "fast-go-icp")

SOURCES = [
    os.path.join(CSRC, "device", "kernels.hip"),
    os.path.join(CSRC, "device", "ctx.hip"),
    os.path.join(CSRC, "device", "bvh.hip"),
    os.path.join(CSRC, "host", "solver.cpp"),
    os.path.join(CSRC, "host", "multi.cpp"),
]
CLI_SOURCES = [os.path.join(CSRC, "cli", "main.cpp")]

# -ffp-contract=off: the arithmetic contract with oracle/ spells out every fma (kernels.hip header)
EXTRA = os.environ.get("FGOICP_EXTRA_CXXFLAGS", "").split()
COMMON_FLAGS = [*EXTRA, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
                "-I" + os.path.join(REPO, "include")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: fgoicp_amd needs ROCm to build")


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    for d in deps:
        for root, _, files in os.walk(d) if os.path.isdir(d) else [(os.path.dirname(d), [], [os.path.basename(d)])]:
            for f in files:
                if os.path.getmtime(os.path.join(root, f)) > t:
                    return False
    return True


def build(force=False, verbose=False, dev=True):
    """Builds the shipped library (and the CLI) and, dev=True, the development build next to it.  Returns the path of the library the
    package loads (FGOICP_LIB or the shipped one)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    deps = [CSRC, os.path.join(REPO, "include")]
    srcs = [s for s in SOURCES if os.path.exists(s)]
    rebuilt = False
    targets = [(DEFAULT_LIB, [])] + ([(DEV_LIB, ["-DFGOICP_DEV_KNOBS"])] if dev else [])
    procs = []
    for path, extra in targets:  # the two builds side by side (kernels.hip dominates either)
        if force or not _newer(path, deps):
            rebuilt = rebuilt or path == DEFAULT_LIB
            cmd = [_hipcc(), "--offload-arch=gfx950", "-x", "hip", *COMMON_FLAGS, *extra, "-shared", "-o", path, *srcs, "-ldl"]  # RCCL is dlopen'ed on first use (csrc/host/multi.cpp)
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cli_srcs = [s for s in CLI_SOURCES if os.path.exists(s)]
    if cli_srcs and (force or rebuilt or not _newer(CLI_PATH, deps)):  # the CLI goes with the library it was built against
        cmd = [_hipcc(), "--offload-arch=gfx950", "-x", "hip", *COMMON_FLAGS, "-o", CLI_PATH, *cli_srcs,
               "-L" + LIB_DIR, "-lfgoicp_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
