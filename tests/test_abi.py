"""The C-ABI shared library loads, exports every symbol include/fgoicp_amd.h declares, and fails
loudly (no CPU fallback) when no GPU is present."""
import os
import re
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(REPO, "include", "fgoicp_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fgoicp_[a-z_0-9]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol(fg):
    path = fg.build.build()
    assert os.path.exists(path)
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r"\b(fgoicp_[a-z_0-9]+)\b", out))
    declared = header_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    assert sorted(fg._lib.exported_symbols()) == declared  # the ctypes table covers the whole header
    lib = fg._lib.load()
    assert b"gfx950" in lib.fgoicp_version()


def test_library_contains_gfx950_code_object(fg):
    path = fg.build.build()
    data = open(path, "rb").read()
    assert b"gfx950" in data and b"bounds_item_kernel" in data


def test_product_does_not_reference_the_oracle(fg):
    """The product must not link, load or import anything under oracle/."""
    path = fg.build.build()
    ldd = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
    assert "oracle" not in ldd
    pkg = os.path.join(REPO, "fast-go-icp_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "pyoracle" not in src and "goicp_oracle" not in src, os.path.join(root, f)


def test_no_gpu_means_loud_failure_not_cpu_fallback(fg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    pts = np.random.default_rng(0).uniform(-1, 1, (16, 3)).astype(np.float32)
    with pytest.raises(fg.FgoicpError) as e:
        fg.Registration(pts, pts, np.array([[-1, 1]] * 3, np.float32), 0.1)
    assert e.value.status == 2 and "no CPU path" in str(e.value)
    with pytest.raises(fg.FgoicpError):
        fg.FastGoICP(pts, pts, 0.1, 1e-3)


def test_missing_library_is_a_loud_error(fg, monkeypatch, tmp_path):
    monkeypatch.setattr(fg._lib, "_lib", None)
    monkeypatch.setattr(fg._lib, "lib_path", lambda: str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fg._lib.load()


def _build_facade_check(tmp_path):
    import subprocess
    exe = str(tmp_path / "facade_check")
    lib_dir = os.path.join(REPO, "fast-go-icp_amd", "lib")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(REPO, "include"),
                    os.path.join(REPO, "tests", "host_harness", "facade_check.cpp"), "-o", exe, "-L" + lib_dir, "-lfgoicp_amd", "-Wl,-rpath," + lib_dir], check=True)
    return exe


def test_cpp_facades_compile_against_the_c_abi_alone(fg, tmp_path):
    """include/fgoicp/*.hpp (the reference's class names over the C ABI) build with a plain C++17 compiler — no HIP headers —
    and, without a GPU, throw the std::runtime_error the reference's CLI path expects instead of computing anything."""
    import subprocess
    exe = _build_facade_check(tmp_path)
    (tmp_path / "pc.txt").write_text("2\n0 0 0\n1 1 1\n")
    (tmp_path / "b.txt").write_text("-1 1 -1 1 -1 1\n")
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run itself is covered by the gpu test")
    p = subprocess.run([exe, str(tmp_path / "pc.txt"), str(tmp_path / "pc.txt"), "0.05", str(tmp_path / "b.txt")], capture_output=True, text=True)
    assert p.returncode != 0 and "no HIP device" in (p.stderr + p.stdout)


def test_invalid_arguments_are_refused_before_any_device_work(fg):
    """NULL pointers, empty device lists and out-of-range ranks come back as FGOICP_ERR_INVALID_ARG (1) from the entry points added in
    round 2 — checked on the CPU: none of these calls reaches the GPU."""
    import ctypes as C
    lib = fg._lib.load()
    assert lib.fgoicp_lut_nodes(None, None, 1, None) == 1
    assert lib.fgoicp_bounds_point_distances(None, None, 0.0, None, 0, None) == 1
    assert lib.fgoicp_ctx_sort_fallbacks(None, None, None) == 1
    assert lib.fgoicp_ctx_profile_select_ms(None, None) == 1
    assert lib.fgoicp_ctx_trim_stats(None, None, 0) == 1
    assert lib.fgoicp_icp_batch(None, 1, None, None, 10, 0.1, None, None, None, None) == 1
    assert lib.fgoicp_bounds_collect(None, 0, None, None) == 1
    assert lib.fgoicp_rccl_unique_id(None) == 1
    h = C.c_void_p()
    ident = (C.c_ubyte * 128)()
    assert lib.fgoicp_rccl_create(0, 1, None, 0, C.byref(h)) == 1 and not h.value
    assert lib.fgoicp_rccl_create(2, 2, ident, 0, C.byref(h)) == 1 and lib.fgoicp_rccl_create(0, 0, ident, 0, C.byref(h)) == 1
    assert lib.fgoicp_rccl_exchange(None, None) == 1 and lib.fgoicp_rccl_calls(None, None) == 1 and lib.fgoicp_rccl_abort(None) == 1
    pts = np.zeros((4, 3), np.float32)
    fp = pts.ctypes.data_as(fg._lib.c_float_p)
    assert lib.fgoicp_multi_create(fp, 4, fp, 4, 0.1, 1e-3, None, None, 0, fg.TRANSPORT_RCCL, C.byref(h)) == 1
    dev = (C.c_int * 1)(0)
    assert lib.fgoicp_multi_create(fp, 4, fp, 4, 0.1, 1e-3, None, dev, 1, 7, C.byref(h)) == 1  # unknown transport
    assert lib.fgoicp_multi_run(None, None, None) == 1 and lib.fgoicp_multi_world(None) == 0 and not lib.fgoicp_multi_solver(None, 0)
    assert lib.fgoicp_multi_recorded(None, 0, None, None) == 1 and lib.fgoicp_multi_set_record(None, 1) == 1 and lib.fgoicp_multi_replay_rank(None, 0, None) == 1 and lib.fgoicp_multi_seconds(None, 0, None) == 1
    assert lib.fgoicp_rccl_comm_count(None, None) == 1 and lib.fgoicp_multi_test_fault(None, 0, 0) == 1 and lib.fgoicp_ctx_test_sort_fault(None, 0) == 1
    lib.fgoicp_multi_destroy(None); lib.fgoicp_rccl_destroy(None)  # no-ops
    assert b"invalid" in lib.fgoicp_last_error() or b"" == lib.fgoicp_last_error()[:0]


def test_cloud_statistics_run_on_the_host(fg):
    """fgoicp_cloud_stats (the reference's open item "compute point clouds' stats", TODO.md:7) needs no device: centroid, box,
    the largest centred coordinate (whose reciprocal over the source is the scale of fgoicp.cpp:205-220) and the RMS radius."""
    import numpy as np
    rng = np.random.default_rng(5)
    p = (rng.normal(size=(5000, 3)) * [1.0, 2.0, 0.5] + [3.0, -1.0, 0.25]).astype(np.float32)
    st = fg.cloud_stats(p)
    c = p.astype(np.float64).mean(0)
    assert st["n"] == 5000 and np.allclose(st["centroid"], c, atol=1e-6)
    assert np.array_equal(st["min"], p.min(0)) and np.array_equal(st["max"], p.max(0))
    assert st["max_abs_centred"] == pytest.approx(float(np.abs(p - c).max()), rel=1e-6)
    assert st["rms_radius"] == pytest.approx(float(np.sqrt(((p - c) ** 2).sum(1).mean())), rel=1e-6)
    assert fg.cloud_stats(np.zeros((0, 3), np.float32))["n"] == 0
    lib = fg._lib.load()
    assert lib.fgoicp_cloud_stats(None, 3, None) == 1 and lib.fgoicp_ctx_get_info(None, None) == 1
    # the driver's pre-processing scales by exactly this statistic of the source
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(p[:3000], p[3000:])
    assert float(scale) == pytest.approx(1.0 / fg.cloud_stats(p[3000:])["max_abs_centred"], rel=2e-6)



def test_abi_revision_and_shipped_build(fg):
    """ABI 2: struct_size leads fgoicp_exchange / fgoicp_ctx_info; the library the package loads by default is the shipped build (no A/B knobs)."""
    import ctypes as C
    import os
    lib = fg._lib.load()
    assert lib.fgoicp_abi_version() == 2
    assert C.sizeof(fg._lib.Exchange) == 48 and fg._lib.Exchange().struct_size == 48
    if not os.environ.get("FGOICP_LIB"):
        assert not fg.dev_knobs()
        # the shipped library carries no knob names (csrc/host/knobs.hpp): what it reads from the environment is FGOICP_HOST_THREADS / _SPIN
        data = open(fg._lib.lib_path(), "rb").read()
        import re
        names = set(re.findall(rb"FGOICP_[A-Z][A-Z_0-9]+", data))
        assert names <= {b"FGOICP_HOST_THREADS", b"FGOICP_HOST_SPIN", b"FGOICP_TRANSPORT_IN_PROCESS", b"FGOICP_BOUNDS_SORTED"}, names
        dev = open(fg.build.DEV_LIB, "rb").read()
        assert len(set(re.findall(rb"FGOICP_[A-Z][A-Z_0-9]+", dev))) > 40
