"""GPU: the CLI end to end (config -> loaders -> solver -> output files) and the torch.distributed
exchange on the nccl (= RCCL) backend."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(REPO, "tests", "golden", "goicp_golden.npz"))


def write_txt(path, pts):
    with open(path, "w") as f:
        f.write(f"{len(pts)}\n")
        for x, y, z in pts:
            f.write(f"{x:.9g} {y:.9g} {z:.9g}\n")


def test_cli_runs_and_matches_library(fg, gpu_required, tmp_path):
    exe = os.path.join(REPO, "fast-go-icp_amd", "lib", "fast-go-icp")
    assert os.path.exists(exe), "CLI not built (python __graft_entry__.py build)"
    tgt, src = G["runsyn_tgt"], G["runsyn_src"]
    write_txt(tmp_path / "tgt.txt", tgt)
    write_txt(tmp_path / "src.txt", src)
    cfg = tmp_path / "cfg.toml"
    cfg.write_text(f'[io]\ntarget = "{tmp_path}/tgt.txt"\nsource = "{tmp_path}/src.txt"\noutput = "{tmp_path}/out.toml"\n'
                   f'visualization = "{tmp_path}/viz.ply"\n[params]\ntarget_subsample = 1.0\nsource_subsample = 0.5\n'
                   f'lut_resolution = {float(G["runsyn_res"])}\nmse_threshold = {float(G["runsyn_mse"])}\nseed = 3\n')
    p = subprocess.run([exe, "-c", str(cfg), "-v"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = p.stdout
    for needle in ("Reading configurations from cfg.toml", "Fast Go-ICP Configurations", "Target point cloud (700) loaded from",
                   "Source point cloud (", "Initial ICP best error:", "Searching over! Best Error:", "Fast Go-ICP finished, time elapsed:"):
        assert needle in out, needle
    # the reference's lines in the reference's order (fgoicp.cpp:15-17, :85-87, :25-27): the initial ICP with its rotation and translation
    # BEFORE the search, a Debug "New best error" block after every refinement the search triggers, then "Searching over!"
    plain = re.sub(r"\x1b\[[0-9;]*m", "", out)
    i0, i1 = plain.index("Initial ICP best error:"), plain.index("Searching over! Best Error:")
    assert re.search(r"Initial ICP best error: [0-9.eE+-]+\n\tRotation:\n(\t[-0-9.]+\t[-0-9.]+\t[-0-9.]+\n){3}\tTranslation: [-0-9.]+\t[-0-9.]+\t[-0-9.]+", plain)
    news = [m.start() for m in re.finditer(r"\[Debug [0-9:]+\] New best error: [0-9.eE+-]+\n\tRotation:\n", plain)]
    assert news and all(i0 < k < i1 for k in news)
    ns = int(re.search(r"Source point cloud \((\d+)\)", out).group(1))
    assert 150 < ns <= 250  # source_subsample clamps to 0.5: floor(500 * 0.5) kept at most
    txt = (tmp_path / "out.toml").read_text()
    sse = float(re.search(r"^sse = (.*)$", txt, re.M).group(1))
    best = float(re.search(r"Best Error: ([0-9.eE+-]+)", out).group(1))
    assert sse == pytest.approx(best, rel=1e-4)
    viz = (tmp_path / "viz.ply").read_text().splitlines()
    assert viz[0] == "ply" and f"element vertex {700 + ns}" in viz
    # the CLI's result against the oracle on the SAME clouds: the loader of the CLI (same file, subsample and seed — the source
    # stream is seeded with seed + 1, main.cpp) through the test harness, then the oracle's FastGoICP (src/main.cpp:41-55)
    import ctypes as C
    from tests.test_cli_host import load as cli_load
    here = os.path.join(REPO, "tests", "host_harness")
    so = os.path.join(here, "libcli_harness.so")
    if not os.path.exists(so):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(here, "cli_harness.cpp")], check=True)
    L = C.CDLL(so)
    L.cli_load_cloud.argtypes = [C.c_char_p, C.c_float, C.c_longlong, C.POINTER(C.c_float), C.c_long, C.c_char_p, C.c_int]
    L.cli_load_cloud.restype = C.c_long
    tgt_l = cli_load(L, tmp_path / "tgt.txt", 1.0, 3)
    src_l = cli_load(L, tmp_path / "src.txt", 0.5, 4)
    assert len(tgt_l) == 700 and len(src_l) == ns
    from oracle import pyoracle
    o = pyoracle.FastGoICP(tgt_l, src_l, float(G["runsyn_res"]), float(G["runsyn_mse"])).run()
    rot = np.array([[float(v) for v in re.findall(r"[-+0-9.eE]+", line)] for line in re.search(r"rotation = \[\n(.*?)\n\]", txt, re.S).group(1).splitlines()], np.float64)
    tr = np.array([float(v) for v in re.findall(r"[-+0-9.eE]+", re.search(r"^translation = \[(.*)\]$", txt, re.M).group(1))])
    assert sse == pytest.approx(float(o["best_sse"]), rel=1e-5) and np.allclose(rot, o["R"], atol=1e-5) and np.allclose(tr, o["t"], atol=1e-5)
    assert len(news) == int(re.search(r"^icp_runs = (\d+)$", txt, re.M).group(1)) - 2  # every ICP but the initial and the final one is a triggered refinement
    st_sub = int(re.search(r"^subcubes = (\d+)$", txt, re.M).group(1))
    assert st_sub == o["stats"]["trans_cubes"]  # the CLI's default schedule follows the reference's exploration order
    # missing config -> usage + non-zero exit; bad extension -> runtime_error (uncaught upstream too)
    assert subprocess.run([exe], capture_output=True).returncode != 0
    # full-cloud run through the CLI equals the library result (same solver underneath)
    cfg.write_text(f'[io]\ntarget = "{tmp_path}/tgt.txt"\nsource = "{tmp_path}/src.txt"\noutput = "{tmp_path}/out2.toml"\n'
                   f'[params]\nsource_subsample = 1.0\nlut_resolution = {float(G["runsyn_res"])}\nmse_threshold = {float(G["runsyn_mse"])}\n')
    # source_subsample is clamped to 0.5 by the reference's Config (utilities.hpp:103), so compare against the library on the same rule
    p = subprocess.run([exe, "-c", str(cfg)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "Source Subsample: 0.5" in p.stdout


def test_nccl_exchange_callbacks(fg, gpu_required):
    """The RCCL transport of the exchange hook (world size 1 here; the multi-rank logic is covered on gloo)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from fgoicp_amd.dist import TorchExchange
        ex = TorchExchange()
        assert ex.device.type == "cuda" and ex.world == 1
        buf = (C.c_float * 3)(3.0, -1.0, 2.5)
        assert ex._allreduce_min(buf, 3, None) == 0 and list(buf) == [3.0, -1.0, 2.5]
        send = (C.c_float * 4)(1, 2, 3, 4)
        recv = (C.c_float * 4)()
        assert ex._allgather(send, recv, 4, None) == 0 and list(recv) == [1, 2, 3, 4]
        # and a solver accepts it
        s = fg.FastGoICP(G["runsyn_tgt"], G["runsyn_src"], float(G["runsyn_res"]), float(G["runsyn_mse"]), schedule=fg.SCHEDULE_ROUND, round_width=2)
        s.set_exchange(ex)
        R, t = s.run()
        assert np.allclose(R, G["runsyn_R"], atol=1e-5)
        s.close()
    finally:
        dist.destroy_process_group()


def _launch_gpu_ranks(tmp_path, world, workload, mse, K):
    import socket
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    prefix = str(tmp_path / f"w{world}")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "tests", "gpu_dist_worker.py"), prefix, workload, repr(mse), str(K)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return [np.load(f"{prefix}.rank{r}.npz") for r in range(world)]


def test_two_ranks_shard_the_search_on_the_gpu(fg, gpu_required, tmp_path):
    """The N > 1 path end to end on the HIP operators: two ranks (sharing this box's one GPU, exchange on gloo) shard every
    round's rotation cubes, keep identical replicated state, and certify the optimum a single rank certifies."""
    tgt, src, R_gt, t_gt = fg.synth.workload("small", angle_deg=150.0, min_angle_deg=110.0)
    one = fg.FastGoICP(tgt, src, 0.02, 2e-5, schedule=fg.SCHEDULE_ROUND, round_width=0)
    R1, t1 = one.run()
    e1, st1 = float(one.get_best_error()), one.stats()
    one.close()
    a, b = _launch_gpu_ranks(tmp_path, 2, "small", 2e-5, 0)
    assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"]) and a["sse"] == b["sse"]
    assert a["rounds"] == b["rounds"] and a["exchange_calls"] == b["exchange_calls"] == 2 * a["rounds"]
    assert float(a["sse"]) == pytest.approx(e1, rel=1e-5) and np.allclose(a["R"], R1, atol=1e-5) and np.allclose(a["t"], t1, atol=1e-5 * max(1.0, float(np.abs(t1).max())))
    # the work is sharded, not replicated
    assert int(a["rot_cubes"]) > 0 and int(b["rot_cubes"]) > 0
    assert abs(int(a["rot_cubes"]) + int(b["rot_cubes"]) - int(st1["rot_cubes"])) <= 0.25 * int(st1["rot_cubes"]) + 16


def test_bench_two_rank_path_rehearsal(gpu_required):
    """bench.py's N > 1 branch (torch.distributed.run, sharded search, max-over-ranks timing, whole-job aggregate), rehearsed
    with both ranks on this box's one GPU and the exchange on gloo: the JSON contract holds and the sharded run evaluates
    exactly the subcubes the single-rank certify run evaluates."""
    import json
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rehearse-on-one-gpu", "--no-dragon", "--no-trimmed"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "strong" and d["unit"] == "subcubes/s" and d["vs_baseline"] is None
    assert d["value"] == pytest.approx(d["subcubes_per_step"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert d["result"]["rotation_error_deg_vs_ground_truth"] < 0.5 and d["reference_default_threshold"]["same_optimum_as_headline"]
    assert 0 < d["rot_cubes_rank0"] < 2236 and "cpu_baseline" not in d  # sharded; the CPU leg is an N = 1 thing
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    # the reference's own trajectory, sharded over the two ranks: the one-GPU SERIAL record of this pair (1 625 992 subcubes, 2 236 rotation cubes)
    ser = d["serial_reference_order"]
    assert ser["subcubes_per_step"] == 1625992 and ser["rot_cubes_rank0"] == 2236 and ser["same_optimum_as_headline"]


def test_cpp_facades_give_the_results_of_the_python_binding(fg, gpu_required, tmp_path):
    """icp::Registration / NearestNeighborLUT / IterativeClosestPoint3D / FastGoICP of include/fgoicp/*.hpp, driven like the
    reference's fgoicp.cpp drives its classes, return bit for bit what the ctypes binding returns for the same calls."""
    import json
    from tests.test_abi import _build_facade_check
    exe = _build_facade_check(tmp_path)
    tgt, src, R_gt, t_gt = fg.synth.workload("tiny", angle_deg=25.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)

    def write(path, pc):
        with open(path, "w") as f:
            f.write(f"{len(pc)}\n")
            for p in pc:
                f.write("%.9g %.9g %.9g\n" % tuple(float(v) for v in p))
    write(tmp_path / "pct.txt", pct); write(tmp_path / "pcs.txt", pcs); write(tmp_path / "tgt.txt", tgt); write(tmp_path / "src.txt", src)
    (tmp_path / "b.txt").write_text(" ".join("%.9g" % float(v) for v in np.asarray(bounds).reshape(-1)) + "\n")
    p = subprocess.run([exe, str(tmp_path / "pct.txt"), str(tmp_path / "pcs.txt"), "0.05", str(tmp_path / "b.txt"), str(tmp_path / "tgt.txt"), str(tmp_path / "src.txt")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    f32 = np.float32
    reg = fg.Registration(pct, pcs, bounds, 0.05)
    assert list(reg.lut_dims()) == d["dims"]
    rn = fg.RotNode(0.25, -0.125, 0.375, 0.125)
    tn = np.array([[0.1 * i - 0.2, 0.05 * i, -0.03 * i, 0.25] for i in range(5)], f32)
    tn[:, :3] = np.array([[f32(0.1) * f32(i) - f32(0.2), f32(0.05) * f32(i), f32(-0.03) * f32(i)] for i in range(5)], f32)
    lb, ub = reg.compute_sse_error(rn, tn, False)
    assert np.array_equal(lb, np.array(d["lb"], f32)) and np.array_equal(ub, np.array(d["ub"], f32))
    t0 = np.array([0.01, -0.02, 0.005], f32)
    assert f32(reg.compute_sse_error(rn.q.R, t0)) == f32(d["sse"])
    icp = fg.IterativeClosestPoint3D(reg, None, None, 100, 0.005, rn.q.R, t0)
    sse, R, t = icp.run()
    assert f32(sse) == f32(d["icp_sse"]) and icp.iterations == d["icp_iters"]
    assert np.array_equal(fg.nodes.to_glm(R), np.array(d["icp_R"], f32)) and np.array_equal(t, np.array(d["icp_t"], f32))
    reg.close()
    s = fg.FastGoICP(tgt, src, 0.05, 1e-3)  # the façade's default: SERIAL
    Rr, tr = s.run()
    assert f32(s.get_best_error()) == f32(d["run_sse"])
    assert np.array_equal(fg.nodes.to_glm(Rr), np.array(d["run_R"], f32)) and np.allclose(tr, np.array(d["run_t"], f32), rtol=0, atol=0)
    s.close()
