#!/usr/bin/env python3
"""Headline benchmark: Go-ICP branch-and-bound on the bunny-shaped cloud (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete FastGoICP::run() (initial ICP, outer SO(3) x inner R^3 branch-and-bound to
the global optimum, final ICP refine — the span the reference's CLI times, src/main.cpp:50-53) on a
seeded synthetic cloud pair that is already resident in HBM (clouds uploaded and LUT built by the
solver constructor, outside the timed region, as in the reference).  value = subcubes evaluated by
all ranks / wall-clock of the K timed steps (max over ranks); a subcube is one (rotation cube,
translation cube) bound evaluation over all ns source points (`count`, fgoicp/fgoicp.cpp:132).

Threshold.  With the reference's default mse_threshold = 1e-3 a clean 40k-point pair stops after ONE
expansion round (8 rotation cubes, 16.5k subcubes): sse_threshold = ns*mse = 40 is far above the
optimum's residual (3.4), so the first good ICP ends the search and the step measures 5 ICP runs, not
the branch-and-bound.  The reference's own example (test/bunny.toml: ~3k source points, ns*mse ~ 3,
below the residual of the shipped clouds) is in the opposite regime: the search has to CERTIFY the
optimum.  The headline step therefore uses mse_threshold = 5e-5 (ns*mse = 2.0 < residual), same
clouds, same LUT, same optimum — ~100x more branch-and-bound work — and the default-threshold run is
reported next to it ("default_threshold_*" keys and "reference_default_threshold").

Legs (every one is a full run() on its own workload; --only LEG runs one alone, e.g. under rocprofv3):
  headline            bunny shape, mse 5e-5, schedule ROUND (adaptive width), K timed steps
  default_threshold   same clouds, mse 1e-3 (BASELINE.md's parameters)
  serial              same clouds, mse 5e-5, the reference's exploration order (SERIAL schedule), one step
  dragon              dragon shape 437 645^2 (configs[2]), mse 5e-6 (certify), one step
  trimmed             1M points, 20 % outliers, trim_fraction 0.2 (configs[4]), mse 1e-3, one step
  cpu_baseline        N = 1 only: the oracle-backed driver (tests/host_harness) on this host's cores

For N > 1 the rotation cubes of every expansion round are sharded over the ranks (one process per
GPU) with one RCCL all-reduce(MIN) of the best error + one small all-gather per round: the total
work is fixed, so "scaling" is "strong".

"roofline" (one object per workload): HIP events around every launch of the bounds kernel on the stream
it runs on (fgoicp_ctx_profile).  `achieved` = algorithmic bytes of the EVALUATIONS the launches did
(ns * (32 + 12/32) B each; a node both tasks of a rotation cube need is two subcubes and one evaluation) / the launches'
duration; `traffic` = HBM bytes per launch from separate rocprofv3 --pmc passes of `--only LEG`
(profiles/bench_pmc.json, see tools/gpu_profile.sh) — `traffic_measured_in_this_run` says so; `hbm_actual_GBps` = traffic / this run's duration.
One `frac` per leg and none above 1: the headline and the trimmed leg are priced against the HBM roof with the measured HBM fraction next
to it; the dense dragon leg, whose per-evaluation byte model is no roof (texels are shared between points and evaluations out of L1 / L2), against the
busiest unit its counters name (`bound`: "ta" texture addresser / "l1" / "valu" / "hbm"; `busiest_unit_fracs` lists the candidates)."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy rate)
LEGS = ("headline", "default_threshold", "serial", "dragon", "trimmed", "cpu_baseline")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="bunny", choices=["bunny", "dragon", "small", "tiny"])
    ap.add_argument("--lut-resolution", type=float, default=0.005)
    ap.add_argument("--mse-threshold", type=float, default=5e-5, help="headline threshold (see module docstring); the reference default 1e-3 is measured as well")
    ap.add_argument("--schedule", default="round", choices=["round", "serial"])
    ap.add_argument("--round-width", type=int, default=0, help="rotation cubes popped per round (0 = adaptive)")
    ap.add_argument("--only", default=None, choices=LEGS, help="run ONE leg (profiling); the headline keys then describe that leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-default-threshold-run", action="store_true")
    ap.add_argument("--no-full-evaluation", action="store_true", help="skip the headline's comparison run with the early exit off")
    ap.add_argument("--no-early-exit", action="store_true", help="A/B: every leg evaluates every subcube in full (fgoicp_solver_set_early_exit(0)); not the product's default")
    ap.add_argument("--no-serial", action="store_true")
    ap.add_argument("--no-dragon", action="store_true", help="skip the secondary dragon-shape (437k points) measurement")
    ap.add_argument("--no-trimmed", action="store_true", help="skip the secondary 1M-point trimmed Go-ICP measurement (20 %% outliers)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="time budget of the one-core operator sample of cpu_baseline")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a one-GPU box: every rank uses device 0 and the exchange runs on gloo (timings are meaningless)")
    return ap.parse_args()


def rot_err_deg(R, R_gt):
    return float(np.degrees(np.arccos(np.clip((np.trace(np.asarray(R, np.float64).T @ R_gt) - 1) / 2, -1, 1))))


class Env:
    """Process-group plumbing shared by the legs."""

    def __init__(self, a):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != a.gpus:
            if self.world == 1 and a.gpus > 1:
                raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
            a.gpus = self.world
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no GPU visible (fgoicp_amd has no CPU path)")
        if a.rehearse_on_one_gpu:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.dist = None
        self.no_early_exit = bool(a.no_early_exit)
        self.red_dev = "cuda"
        self.ex = None
        self.transport = None
        self.comm_ranks = None
        if self.world > 1:
            # The trimmed leg's cooperative refinements would split their scans over the ranks through an in-place RCCL all-gather on device
            # memory (fgoicp_exchange.allgather_device) — a collective no multi-GPU node has executed yet (tests: in-process ranks on one
            # device, gloo).  The scaling run's headline must not depend on it: refinements run replicated here unless the caller opts in
            # (FGOICP_BENCH_SPLIT_SCANS=1 keeps the library's default: trimmed contexts of at least 262144 points split; fgoicp_ctx_set_coop_split).
            self.split_scans = os.environ.get("FGOICP_BENCH_SPLIT_SCANS", "0") != "0"  # opt in: run_leg() then leaves the library's default (trimmed contexts split)
            import torch.distributed as dist
            self.dist = dist
            if a.rehearse_on_one_gpu:
                dist.init_process_group("gloo")
                self.red_dev = "cpu"
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            if a.rehearse_on_one_gpu:
                from fgoicp_amd.dist import TorchExchange
                self.ex = TorchExchange()
                self.transport = "torch.distributed gloo (rehearsal)"
            else:  # the library's own RCCL transport (no Python in the per-round collectives); the id travels over torch.distributed
                import fgoicp_amd as fg
                self.transport = "rccl inside libfgoicp_amd.so (fgoicp_rccl_*)"
                try:
                    ident = torch.zeros(128, dtype=torch.uint8, device="cuda")
                    if self.rank == 0:
                        ident.copy_(torch.frombuffer(bytearray(fg.rccl_unique_id()), dtype=torch.uint8))
                    dist.broadcast(ident, 0)
                    self.ex = fg.RcclExchange(self.rank, self.world, bytes(ident.cpu().numpy().tobytes()), self.local_rank)
                    ok = torch.tensor([1.0], device="cuda")
                except Exception as e:  # keep the run alive on the torch.distributed transport (ctypes callbacks) and say so
                    print(f"[bench] rank {self.rank}: in-library RCCL exchange unavailable ({e!r})", file=sys.stderr, flush=True)
                    ok = torch.tensor([0.0], device="cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if float(ok) < 1.0:  # every rank must use the same transport
                    from fgoicp_amd.dist import TorchExchange
                    self.ex = TorchExchange()
                    self.transport = "torch.distributed nccl through ctypes callbacks (fallback)"
            self.ex.warmup()  # communicator setup is not part of a registration run
            try:  # what the communicator itself says about its size (ncclCommCount): a scaling record shows whether RCCL really joined N ranks
                self.comm_ranks = int(self.ex.comm_count) if hasattr(self.ex, "comm_count") else None
            except Exception:
                self.comm_ranks = None

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def sum_max(self, values):
        """-> (sum over ranks, max over ranks) of a list of floats"""
        t = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=self.red_dev)
        if self.dist is None:
            return t.tolist(), t.tolist()
        m = t.clone()
        self.dist.all_reduce(m, op=self.dist.ReduceOp.MAX)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.tolist(), m.tolist()


def run_leg(env, fg, tgt, src, res, mse, sched, K, steps, warmup, trim=0.0, early_exit=True):
    """W warm-up + exactly `steps` timed run()s bracketed by barrier + synchronize; wall = max over ranks, subcubes = sum.
    early_exit (the library's default): the inner BnBs hand their thresholds to the bounds operator, which stops evaluating a subcube
    the search drops anyway (fgoicp_bounds_submit_cut; same trajectory, counters and result as evaluating everything in full)."""
    t0 = time.perf_counter()
    solver = fg.FastGoICP(tgt, src, res, mse, schedule=sched, round_width=K, device=env.local_rank, trim_fraction=trim)
    env.torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    if env.ex is not None:
        solver.set_exchange(env.ex)
    solver.set_early_exit(early_exit and not env.no_early_exit)
    reg = solver.registration
    if env.world > 1 and not getattr(env, "split_scans", False):
        reg.set_coop_split(None, None)  # N > 1: refinements replicated on every rank (see Env)
    for _ in range(warmup):
        solver.run()
    reg.set_profile(True)  # HIP events around every bounds kernel of the timed region (~2 % of a step)
    reg.profile(reset=True)
    reg.cut_stats(reset=True)
    sub, stats, R, t = 0, None, None, None
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        R, t = solver.run()
        stats = solver.stats()
        sub += stats["trans_cubes"]
    env.barrier()
    elapsed = time.perf_counter() - t0
    prof = reg.profile(reset=True)
    prof["work_items_offered"], prof["work_items_not_evaluated"] = reg.cut_stats(reset=True)
    reg.set_profile(False)
    sums, maxes = env.sum_max([sub, elapsed])
    if stats is not None and steps > 1:  # ICP figures over all timed steps (stats() describes the last run only)
        stats = dict(stats)
    info = reg.info()
    prof["chunks_per_evaluation"] = info["items_per_evaluation"]
    prof["chunks_per_work_item"] = info["chunks_per_item_with_thresholds"] if (early_exit and not env.no_early_exit and trim == 0.0) else 1
    out = dict(solver=solver, reg=reg, R=R, t=t, stats=stats, prof=prof, elapsed=maxes[1], subcubes=sums[0], subcubes_rank=sub, steps=steps, setup_s=setup_s, trim=trim,
               best_sse=float(solver.get_best_error()), ns=reg.ns, nt=reg.nt, lut_dims=list(reg.lut_dims()))
    return out


def unit_bytes(ns):
    return ns * (32.0 + 12.0 / 32.0)  # per subcube, batches of B = 32 sharing one rotated source


def roofline(leg, pmc, extra=None):
    """The bounds kernel of one leg (rank 0's launches) against the HBM roof."""
    p, ns = leg["prof"], leg["ns"]
    launches, kms = p["launches"], p["kernel_ms"]
    if not launches or kms <= 0:
        return None
    ub = unit_bytes(ns)
    # early exit: the kernel is priced on the work items (evaluation x chunk of source points) it EVALUATED, counted on the device
    offered, skipped = p.get("work_items_offered", 0), p.get("work_items_not_evaluated", 0)
    done_frac = 1.0 - skipped / offered if offered else 1.0
    ach = p["evaluations"] * done_frac * ub / (kms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": "bounds_item_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": None, "traffic_measured_in_this_run": False,
         "avg_launch_us": kms * 1e3 / launches, "launches": int(launches), "evaluations_per_launch": p["evaluations"] / launches,
         "subcubes_per_launch": leg["subcubes_rank"] / launches, "output_rows_per_launch": p["subcubes"] / launches, "algorithmic_bytes_per_evaluation": ub,
         "algorithmic_bytes_per_launch": p["evaluations"] * done_frac * ub / launches,
         "work_items_evaluated_frac": done_frac,
         "early_exit": bool(offered),
         "subcubes_served_per_evaluation": leg["subcubes_rank"] / max(1.0, float(p["evaluations"])),
         "note": "achieved = algorithmic bytes of the EVALUATIONS (SURVEY 8d unit x evaluations per launch x the fraction of their work items the kernel evaluated: a subcube "
                 "whose lower bound has reached the threshold its inner BnB drops it at is not evaluated further, fgoicp_bounds_submit_cut; counted on the device) / launch duration, "
                 "HIP events on the kernel's own stream; a "
                 "translation node that both the UB and the LB task of a rotation cube need is evaluated once for both (same tick: twin; later: the LB task takes it from "
                 "its memo) — subcubes_served_per_evaluation; output rows include the memo's look-ahead rows.  `traffic` and everything under `utilisation` are per-launch "
                 "counter figures of separate rocprofv3 --pmc passes of the same deterministic step on this tree (profiles/bench_pmc.json, bench_pmc_extra.json), NOT measured "
                 "in this run; rates derived from them use this run's launch duration"}
    # the launch against the two floors it is made of when subcubes end early: the evaluated work items at the rate the kernel reaches when it
    # evaluates everything (the same tree's full-evaluation figure is in the line), and one workgroup dispatch per work item
    # (tools/calib/dispatch_rate.hip, profiles/r04_dispatch_rate.txt: 0.212 ns per 64-thread workgroup, whatever it does)
    span = max(1, int(p.get("chunks_per_work_item", 1)))
    wgs = p["evaluations"] / launches * (-(-int(p.get("chunks_per_evaluation", 0)) // span)) if p.get("chunks_per_evaluation") else None
    if wgs:
        r["workgroups_per_launch"] = wgs
        r["workgroup_dispatch_floor_us"] = wgs * 0.212e-3
    if pmc:
        r["traffic"] = pmc.get("hbm_bytes_per_launch")
        r["traffic_source"] = pmc.get("source")
        r["traffic_read_bytes_per_launch"] = pmc.get("read_bytes_per_launch")
        r["traffic_write_bytes_per_launch"] = pmc.get("write_bytes_per_launch")
        r["l2_hit_rate"] = pmc.get("l2_hit_rate")
        if pmc.get("hbm_bytes_per_launch"):  # the passes run the same step on the same tree: bytes per launch carry over, durations are this run's
            act = pmc["hbm_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9
            if act <= HBM_PEAK_GBS:
                r["hbm_actual_GBps"] = act
                r["hbm_actual_frac"] = act / HBM_PEAK_GBS
                r["traffic_over_algorithmic"] = pmc["hbm_bytes_per_launch"] / r["algorithmic_bytes_per_launch"]
            else:  # bytes of a slower kernel over this one's duration: the counter file is of another tree — no rate is derived from it
                r["traffic_stale"] = "profiles/bench_pmc.json was recorded on a tree whose launches move more bytes than this run's duration allows: re-run tools/gpu_profile.sh"
        if pmc.get("limited_by"):
            r["limited_by"] = pmc["limited_by"]
    if extra:
        r.update(extra)
    return r


VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0  # G wave64-instructions/s: 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles, 2.4 GHz max clock


def utilisation(r, x):
    """What binds the bounds kernel besides HBM bytes: the units the kernel keeps busy, from separate rocprofv3 --pmc passes
    of the same deterministic step (profiles/bench_pmc_extra.json), each as achieved / peak <= 1.  The instruction count per launch carries
    over to this run (same step, same code); rates use THIS run's launch duration, the cycle-based fractions are the profiled pass's own."""
    if not r or not x:
        return
    dur = r["avg_launch_us"] * 1e-6
    u = {"source": x.get("source")}
    if x.get("valu_insts_per_launch"):
        ach = x["valu_insts_per_launch"] / dur / 1e9
        u["valu"] = {"achieved": ach, "peak": VALU_PEAK_GINST, "unit": "G wave64 VALU instructions/s", "frac": ach / VALU_PEAK_GINST,
                     "frac_cycle_based": x.get("valu_issue_utilisation"), "insts_per_launch": x["valu_insts_per_launch"],
                     "insts_per_point_evaluation": x["valu_insts_per_launch"] * 64.0 / (r["evaluations_per_launch"] * r.get("work_items_evaluated_frac", 1.0) * r["algorithmic_bytes_per_evaluation"] / 32.375)}
    for k in ("l1_hit_rate", "l1_miss_latency_cycles", "l1_pending_stall_frac", "l1_accesses_per_clock_cu", "ta_busy_frac", "ta_addr_stalled_frac", "ta_data_stalled_frac", "wave_wait_frac",
              "wave_issue_stall_frac"):
        if x.get(k) is not None:
            u[k] = x[k]
    if x.get("l1_accesses_per_launch"):
        u["l1_accesses_per_point_evaluation"] = x["l1_accesses_per_launch"] / (r["evaluations_per_launch"] * r.get("work_items_evaluated_frac", 1.0) * r["algorithmic_bytes_per_evaluation"] / 32.375)
    r["utilisation"] = u
    # every candidate is a fraction of something that cannot exceed 1: HBM bytes moved / peak, VALU issue cycles / cycles, texture-addresser busy cycles / cycles,
    # L1 cache-line accesses per clock and CU (the TCP looks up one line per clock)
    cands = {"hbm": r.get("hbm_actual_frac") or 0.0, "valu": (u.get("valu") or {}).get("frac_cycle_based") or (u.get("valu") or {}).get("frac") or 0.0, "ta": u.get("ta_busy_frac") or 0.0,
             "l1": u.get("l1_accesses_per_clock_cu") or 0.0}
    r["busiest_unit_fracs"] = cands
    r["busiest_unit"] = max(cands, key=cands.get)


def cpu_baseline(fg, reg, tgt, src, res, mse, sched_id, K, seconds, gpu_leg):
    """The reference has no CPU path, so the baseline is the product's host driver over the CPU oracle's operators
    (tests/host_harness: same driver template, thresholds and LUT semantics) with a uniform-grid exact nearest-neighbour search
    standing in for the nanoflann kd-tree the reference's README names.  A full run() to the optimum at the reference's default
    threshold on the host's cores (kind "port"); its LUT is filled from the device LUT (bit-identical; the O(nodes * nt) CPU
    build would take hours and is outside the timed span on the GPU side as well).  Plus the bounds operator on ONE core for a
    bounded sample."""
    from oracle import pyoracle
    from tests import host_harness as hh
    os.environ["FGOICP_HOST_SPIN"] = "0"     # the CPU run's driver shares the cores with the oracle's OpenMP team, which does the parallel work:
    os.environ["FGOICP_HOST_THREADS"] = "1"  # no polling worker threads next to it
    pyoracle.build()
    cores = int(pyoracle.lib().orc_num_threads())
    h = hh.HostDriver(tgt, src, res, mse, schedule=sched_id, round_width=K, build_lut=False, use_grid=True)
    assert h.lut_dims() == tuple(reg.lut_dims())
    h.lut_set(reg.lut_read())
    t0 = time.perf_counter()
    r = h.run()
    wall = time.perf_counter() - t0
    secs = h.seconds()
    same = bool(abs(float(r["best_sse"]) - gpu_leg["best_sse"]) <= 1e-5 * gpu_leg["best_sse"] and np.allclose(r["R"], gpu_leg["R"], atol=1e-5))
    # the bounds operator on ONE core, bounded sample of the same workload
    pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
    orc = pyoracle.Registration(pct, pcs, bounds, res, build_lut=False)
    orc.lut_set(reg.lut_read())
    rng = np.random.default_rng(0)
    rn = fg.RotNode(0.25, -0.125, 0.375, 0.125)
    pyoracle.lib().orc_set_num_threads(1)
    done1, t1 = 0, time.perf_counter()
    while True:
        tn = np.concatenate([rng.uniform(-0.5, 0.5, (32, 3)), np.full((32, 1), 0.125)], axis=1).astype(np.float32)
        orc.compute_bounds(rn.q.R, rn.span, tn, False)
        done1 += 32
        dt1 = time.perf_counter() - t1
        if dt1 >= seconds:
            break
    pyoracle.lib().orc_set_num_threads(cores)
    sub = r["stats"]["trans_cubes"]
    return {"value": sub / wall, "unit": "subcubes/s", "cores": cores, "kind": "port",
            "sample": f"one full run() to the optimum: bunny-shape pair, mse_threshold={mse} (the reference's default), {sub} subcubes, "
                      f"{r['stats']['icp_runs']} ICP runs ({r['stats']['icp_iters']} iterations), OpenMP over points/queries on {cores} threads",
            "wall_clock_to_optimum_s": wall, "seconds_bnb": secs["bnb"], "seconds_icp": secs["icp"], "subcubes": int(sub),
            "same_optimum_as_gpu": same, "best_sse": float(r["best_sse"]),
            "gpu_wall_clock_to_optimum_s_same_run": gpu_leg["elapsed"] / gpu_leg["steps"],
            "value_1_core": done1 / dt1, "sample_1_core": f"bounds operator only, {done1} subcubes (batches of 32, fix_rot=0) in {dt1:.1f}s on one thread",
            "nearest_neighbour": "uniform grid over the target (oracle/goicp_oracle.cpp GridNN; exact, identical to the O(n*m) loops)"}


def icp_latency(leg):
    """The ICP path of a leg (IterativeClosestPoint3D::run, icp3d.cu:80-108; exact SSE registration.cu:62-86) is a chain of dependent
    passes, not a bandwidth kernel: per iteration one correspondence scan of the working cloud (+ the move of the cloud and the
    wave-level sums of points and correspondences in its epilogue), one covariance pass (+ centroids), one exact-SSE scan of the
    pristine cloud next to them on a second stream, two host syncs (3x3 SVD; loop test).  Reported as a latency object: what an
    iteration costs against what its bytes would cost at the HBM roof."""
    st = leg["stats"]
    it, runs, sec = int(st["icp_iters"]), int(st["icp_runs"]), float(st["seconds_icp"])
    if it <= 0:
        return None
    ns, nt = leg["ns"], leg["nt"]
    fused = ns <= 262144 and not leg.get("trim")
    # bytes one iteration has to move at least: correspondence pass reads the working cloud (16 B) and the seed index (4), writes the
    # moved cloud (16) and the index (4); covariance pass reads cloud (16), index (4) and the correspondence (16); SSE pass reads the
    # pristine cloud (16) and the seed (4), writes the minima (4); both scans walk the target once at least (2 x 16 B per target point)
    bytes_it = ns * (16 + 4 + 16 + 4 + 16 + 4 + 16 + 16 + 4 + 4) + 2 * nt * 16
    us = sec / it * 1e6
    floor_us = bytes_it / (HBM_PEAK_GBS * 1e9) * 1e6
    kern = ("nn_scan_kernel<1> -> icp_cov_cen_kernel | nn_scan_kernel<0> (two streams)" if fused else
            "nn_scan_dual_kernel (one walk for both scans) -> " + ("icp_inliers, " if leg.get("trim") else "") + "icp_sums, icp_centroids, icp_cov, sums")
    return {"bound": "latency", "kernel": kern, "iterations": it, "icp_runs": runs, "seconds_icp": sec,
            "us_per_iteration": us, "launches_per_iteration": 3 if fused else 9, "host_syncs_per_iteration": 2,
            "algorithmic_bytes_per_iteration": bytes_it, "hbm_floor_us_per_iteration": floor_us, "frac_of_hbm_floor": floor_us / us,
            "note": "seconds_icp / iterations over the leg's timed steps (every ICP run of FastGoICP::run: initial, triggered, final); iterations far from "
                    "convergence scan many target leaves (the exact search must visit every leaf closer than the current best), converged ones few: "
                    "tools/icp_bench.py isolates one run (profiles/r03_icp_*)"}


def leg_summary(leg, R_gt, t_gt, what):
    st = leg["stats"]
    return {"workload": what, "subcubes_per_s": leg["subcubes"] / leg["elapsed"], "wall_clock_to_optimum_s": leg["elapsed"] / leg["steps"],
            "subcubes_per_step": leg["subcubes"] / leg["steps"], "rot_cubes_rank0": st["rot_cubes"], "icp_runs_rank0": st["icp_runs"], "rounds": st["rounds"],
            "seconds_bnb_rank0": st["seconds_bnb"], "seconds_icp_rank0": st["seconds_icp"], "best_sse": leg["best_sse"],
            "bnb_without_icp_subcubes_per_s_rank0": st["trans_cubes"] / (st["seconds_total"] - st["seconds_icp"]) if st["seconds_total"] > st["seconds_icp"] else None,
            "rotation_error_deg_vs_ground_truth": rot_err_deg(leg["R"], R_gt), "translation_error_vs_ground_truth": float(np.linalg.norm(leg["t"] - t_gt)),
            "setup_s_upload_plus_lut_build": leg["setup_s"], "lut_dims": leg["lut_dims"]}


def main():
    a = parse()
    import fgoicp_amd as fg
    env = Env(a)
    world, rank = env.world, env.rank
    want = lambda name: (a.only is None or a.only == name)
    pmc_all, pmc_extra = {}, {}
    try:
        pmc_all = json.load(open(os.path.join(REPO, "profiles", "bench_pmc.json")))
    except Exception:
        pass
    try:  # SQ / TA / TCP passes of the same legs (tools/pmc_extra.sh -> tools/pmc_extra_summary.py)
        pmc_extra = json.load(open(os.path.join(REPO, "profiles", "bench_pmc_extra.json")))
    except Exception:
        pass

    K = a.round_width
    sched = fg.SCHEDULE_ROUND if a.schedule == "round" else fg.SCHEDULE_SERIAL
    tgt, src, R_gt, t_gt = fg.synth.workload(a.workload, angle_deg=150.0, min_angle_deg=110.0)  # rotation far outside the ICP basin
    line = {"metric": "BnB subcubes/sec + wall-clock to global optimum, bunny 40k pts, 1/2/4/8 GPU", "value": None, "unit": "subcubes/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic"}
    head = None
    if want("headline"):
        head = run_leg(env, fg, tgt, src, a.lut_resolution, a.mse_threshold, sched, K, a.steps, a.warmup)
        line.update({"value": head["subcubes"] / head["elapsed"], "ms_per_step": head["elapsed"] / a.steps * 1e3,
                     "config": {"workload": f"{a.workload}-shape synthetic pair (nt={len(tgt)}, ns={len(src)}), lut_resolution={a.lut_resolution}, "
                                            f"mse_threshold={a.mse_threshold}, full FastGoICP::run() per step",
                                "schedule": a.schedule, "round_width": K if K > 0 else "adaptive (32 per rank, doubled after each round that leaves the incumbent standing)",
                                "lut_dims": head["lut_dims"], "transport": env.transport, "rccl_comm_count": env.comm_ranks, "parallelism": f"rotation cubes sharded over {world} rank(s), one allreduce(min) + one allgather per round" + (f"; transport: {env.transport}" if env.transport else "")},
                     "wall_clock_to_optimum_s": head["elapsed"] / a.steps, "subcubes_per_step": head["subcubes"] / a.steps})
        s = leg_summary(head, R_gt, t_gt, "headline")
        line.update({k: s[k] for k in ("rot_cubes_rank0", "icp_runs_rank0", "rounds", "seconds_bnb_rank0", "seconds_icp_rank0", "setup_s_upload_plus_lut_build")})
        line["result"] = {"best_sse": head["best_sse"], "rotation_error_deg_vs_ground_truth": s["rotation_error_deg_vs_ground_truth"],
                          "translation_error_vs_ground_truth": s["translation_error_vs_ground_truth"]}
        line["roofline"] = roofline(head, pmc_all.get("headline"), {})
        if a.only in (None, "headline") and not a.no_full_evaluation and world == 1:
            # the same steps with every subcube evaluated in full, as the reference evaluates them (fgoicp_solver_set_early_exit(0)): the
            # same trajectory, counters and incumbent — checked here — at the price of the point evaluations the search never looks at
            full = run_leg(env, fg, tgt, src, a.lut_resolution, a.mse_threshold, sched, K, a.steps, 1, early_exit=False)
            rf = roofline(full, None)
            line["full_evaluation"] = {
                "what": "the headline steps with the early exit off: every subcube evaluated in full (the reference's kernComputeBounds does; same search, same result)",
                "value": full["subcubes"] / full["elapsed"], "unit": "subcubes/s", "ms_per_step": full["elapsed"] / a.steps * 1e3,
                "same_counters_as_headline": bool(all(full["stats"][k] == head["stats"][k] for k in ("trans_cubes", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb", "rounds"))),
                "same_incumbent_bits_as_headline": bool(np.array_equal(full["R"], head["R"]) and np.array_equal(full["t"], head["t"]) and full["best_sse"] == head["best_sse"]),
                "roofline": None if rf is None else {k: rf[k] for k in ("achieved", "peak", "unit", "frac", "avg_launch_us", "launches", "evaluations_per_launch", "work_items_evaluated_frac")}}
            full["solver"].close()
            line["value_every_subcube_evaluated_in_full"] = line["full_evaluation"]["value"]
        line["early_exit"] = {
            "on": bool(line["roofline"] and line["roofline"].get("early_exit")),
            "what": "`value` is measured with the library's default: every inner branch-and-bound tells the bounds kernel the value above which it drops a subcube whatever its exact "
                    "bounds are (fgoicp.cpp:151; the results it passes on are only ever compared: :74, :92), and the kernel stops evaluating a subcube once the lower-bound sums "
                    "(all terms >= 0) of its finished work items have reached that value.  No subcube is skipped and none is answered approximately where the search looks: "
                    "trajectory, counters and incumbent are those of the full evaluation, bit for bit (tests/test_host_logic.py, tests/test_gpu_fullsize.py; on one GPU also "
                    "checked in this run: full_evaluation.same_counters_as_headline / same_incumbent_bits_as_headline, with `value_every_subcube_evaluated_in_full` = the same "
                    "steps under fgoicp_solver_set_early_exit(0), as the reference's kernComputeBounds evaluates).  `roofline` prices the kernel on the work items it evaluated.",
            "work_items_evaluated_frac": line["roofline"].get("work_items_evaluated_frac") if line["roofline"] else None}
        utilisation(line["roofline"], pmc_extra.get("headline"))
        u = line["roofline"].get("utilisation") or {}
        if u.get("valu"):  # the counters of THIS kernel build (profiles/bench_pmc_extra.json [headline]), not a remembered figure
            r = line["roofline"]
            fe = (line.get("full_evaluation") or {}).get("roofline") or {}
            floors = ""
            if fe.get("avg_launch_us") and r.get("workgroup_dispatch_floor_us"):
                floors = (f"; a launch = {r['avg_launch_us']:.0f} us against {r['work_items_evaluated_frac'] * fe['avg_launch_us']:.0f} us for the evaluated work items at the full-evaluation "
                          f"kernel's rate (this run: {fe['avg_launch_us']:.0f} us, {fe['frac']:.2f} of the HBM roof) and a floor of {r['workgroup_dispatch_floor_us']:.0f} us for dispatching its "
                          f"{r['workgroups_per_launch']:.0f} workgroups — the ones that end early cost little else")
            r["limited_by"] = (
                f"evaluated work items: L1-miss concurrency x L2 latency, as without the early exit; items that end early: the workgroup dispatch rate (counters of this kernel, "
                f"profiles/bench_pmc_extra.json [headline]: L1 hit rate {100 * u.get('l1_hit_rate', 0):.0f} %, the L1 in pending-stall {100 * u.get('l1_pending_stall_frac', 0):.0f} % of its cycles, "
                f"{u.get('l1_miss_latency_cycles', 0):.0f} cycles per L1 miss, VALU issue {100 * (u['valu'].get('frac_cycle_based') or u['valu']['frac']):.0f} %, TA busy {100 * (u.get('ta_busy_frac') or 0):.0f} %, "
                f"measured HBM traffic {100 * (r.get('hbm_actual_frac') or 0):.0f} % of the peak){floors}")

    # BASELINE.md's parameters (mse_threshold 1e-3) on the same clouds
    dflt = None
    if want("default_threshold") and not a.no_default_threshold_run:
        dflt = run_leg(env, fg, tgt, src, a.lut_resolution, 1e-3, sched, K, 3, 1)
        s = leg_summary(dflt, R_gt, t_gt, f"{a.workload}-shape pair, mse_threshold=0.001 (BASELINE.md's parameters), 3 steps after 1 warm-up")
        s["mse_threshold"] = 1e-3
        if head is not None:
            s["same_optimum_as_headline"] = bool(np.allclose(dflt["R"], head["R"], atol=1e-5) and np.allclose(dflt["t"], head["t"], atol=1e-5 * max(1.0, float(np.abs(head["t"]).max()))))
        line["reference_default_threshold"] = s
        line["default_threshold_ms_per_step"] = s["wall_clock_to_optimum_s"] * 1e3
        line["default_threshold_subcubes_per_s"] = s["subcubes_per_s"]
        line["default_threshold_subcubes_per_step"] = s["subcubes_per_step"]
        line["icp_latency"] = icp_latency(dflt)

    # Secondary legs.  On N > 1 ranks a leg that fails (a collective on a node nobody has run on before) must not cost the run its headline:
    # the failure is recorded in the line, the exchange is aborted for every rank (the peers' runs then fail fast instead of waiting) and the
    # remaining secondary legs are skipped.  On one rank an exception propagates as before.
    state = {"failed": None}

    def guarded(name, fn):
        if state["failed"]:
            line.setdefault("skipped_after_failure", []).append(name)
            return
        if world == 1:
            fn()
            return
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            state["failed"] = name
            line[name + "_error"] = repr(e)[:500]
            print(f"[bench] rank {rank}: leg {name} failed: {e!r}", file=sys.stderr, flush=True)
            try:
                if env.ex is not None and hasattr(env.ex, "abort"):
                    env.ex.abort()
            except Exception:
                pass

    # the reference's own exploration order (the drop-in classes' default schedule)
    if want("serial") and not a.no_serial:
        def _leg_serial():
            ser = run_leg(env, fg, tgt, src, a.lut_resolution, a.mse_threshold, fg.SCHEDULE_SERIAL, 1, 1, 1)
            if world > 1:  # SERIAL on N ranks: the trajectory is replicated, every rank's counters are the whole run's — not to be summed
                ser["subcubes"] = ser["subcubes"] / world
            s = leg_summary(ser, R_gt, t_gt, f"{a.workload}-shape pair, mse_threshold={a.mse_threshold}, SERIAL schedule (the reference's pops, pushes and counters — "
                                              "checked against the oracle's literal driver in tests), one step after 1 warm-up"
                                              + (f"; the inner BnBs of every speculative evaluation dealt over {world} ranks, one all-gather per evaluation" if world > 1 else ""))
            if head is not None:
                s["same_optimum_as_headline"] = bool(np.allclose(ser["R"], head["R"], atol=1e-5) and abs(ser["best_sse"] - head["best_sse"]) <= 1e-5 * head["best_sse"])
            s["roofline"] = roofline(ser, None)
            line["serial_reference_order"] = s
            ser["solver"].close()
        guarded("serial", _leg_serial)

    # dragon shape (BASELINE configs[2]/[3]; nt = ns = 437 645), certify regime, one step
    if want("dragon") and not a.no_dragon and a.workload == "bunny":
        def _leg_dragon():
            tgt_d, src_d, R_gt_d, t_gt_d = fg.synth.workload("dragon", angle_deg=150.0, min_angle_deg=110.0)
            dr = run_leg(env, fg, tgt_d, src_d, a.lut_resolution, 5e-6, sched, K, 1, 0)  # ns*mse = 2.2 < residual 3.0
            s = leg_summary(dr, R_gt_d, t_gt_d, f"dragon-shape synthetic pair (nt={len(tgt_d)}, ns={len(src_d)}), mse_threshold=5e-06, one step, no warm-up")
            r = None
            if rank == 0:
                r = roofline(dr, pmc_all.get("dragon"), {})
                if r:
                    utilisation(r, pmc_extra.get("dragon"))
                    u = r.get("utilisation") or {}
                    # Dense cloud: neighbouring points share texels and the LUT lines are re-used ACROSS the evaluations of a tick out of L1 / L2, so the per-evaluation
                    # byte model of SURVEY 8d (8 private texels per point) is not a roof here (priced that way the kernel would "exceed" the HBM peak).  One `frac`: the
                    # busiest unit the counters of this kernel name (profiles/bench_pmc_extra.json [dragon]); the HBM side as what it is, measured traffic / duration.
                    r["algorithmic_model_GBps_not_a_roof"] = r["achieved"]
                    if r.get("hbm_actual_GBps"):
                        r["hbm"] = {"achieved": r["hbm_actual_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["hbm_actual_frac"],
                                    "definition": "measured HBM bytes per launch (rocprofv3 PMC, FETCH_SIZE x 2 + WRITE_SIZE; not measured in this run) / this run's launch duration"}
                    b = r.get("busiest_unit")
                    if b == "ta" and u.get("ta_busy_frac"):
                        r["bound"], r["achieved"], r["peak"], r["unit"], r["frac"] = "ta", u["ta_busy_frac"], 1.0, "texture-addresser busy cycles / cycle (TA_BUSY, average over the 256 TAs)", u["ta_busy_frac"]
                    elif b == "l1" and u.get("l1_accesses_per_clock_cu"):
                        r["bound"], r["achieved"], r["peak"], r["unit"], r["frac"] = "l1", u["l1_accesses_per_clock_cu"], 1.0, "L1 cache-line accesses per clock and CU (TCP_TOTAL_CACHE_ACCESSES)", u["l1_accesses_per_clock_cu"]
                    elif b == "valu" and u.get("valu"):
                        v = u["valu"]
                        r["bound"], r["achieved"], r["peak"], r["unit"], r["frac"] = "valu", v["achieved"], v["peak"], v["unit"], v.get("frac_cycle_based") or v["frac"]
                    elif r.get("hbm_actual_GBps"):
                        r["achieved"], r["frac"] = r["hbm_actual_GBps"], r["hbm_actual_frac"]
                    else:  # no counter file for this tree: nothing to price against — say so instead of printing a model above the peak
                        r["bound"], r["achieved"], r["frac"] = "unpriced (no counter passes of this tree)", None, None
                    if u:
                        r["limited_by"] = (f"the L1 / texture path: TA busy {100 * (u.get('ta_busy_frac') or 0):.0f} % of the cycles ({100 * (u.get('ta_addr_stalled_frac') or 0):.0f} % stalled by the L1), "
                                           f"{(u.get('l1_accesses_per_clock_cu') or 0):.2f} cache-line accesses per clock and CU ({(u.get('l1_accesses_per_point_evaluation') or 0):.2f} per point-evaluation, hit rate "
                                           f"{100 * (u.get('l1_hit_rate') or 0):.0f} %), VALU issue {100 * ((u.get('valu') or {}).get('frac_cycle_based') or 0):.0f} % "
                                           f"({(u.get('valu') or {}).get('insts_per_point_evaluation', 0):.0f} VALU instructions per point-evaluation: round 4 cut them from 120 to 77 with packed fp32 and the launch "
                                           f"did not get shorter — profiles/r04_ab_item_kernel_dragon_trimmed.txt), measured HBM traffic {100 * (r.get('hbm_actual_frac') or 0):.0f} % of the peak")
                s["roofline"] = r
            s["icp_latency"] = icp_latency(dr)
            line["dragon_shape"] = s
            dr["solver"].close()
        guarded("dragon", _leg_dragon)

    # BASELINE configs[4] — 1M points, 20 % uniform outliers, trimmed Go-ICP (an extension: the reference parses `trim` and ignores it)
    if want("trimmed") and not a.no_trimmed and a.workload == "bunny":
        def _leg_trimmed():
            tgt_m, src_m, R_gt_m, t_gt_m = fg.synth.workload("synthetic1m_outliers", angle_deg=150.0, min_angle_deg=110.0)
            tr = run_leg(env, fg, tgt_m, src_m, a.lut_resolution, 1e-3, sched, K, 1, 0, trim=0.2)
            s = leg_summary(tr, R_gt_m, t_gt_m, f"1M-point synthetic pair, 20 % of the source replaced by uniform outliers (nt={len(tgt_m)}, ns={len(src_m)}), trim_fraction=0.2, "
                                                 "mse_threshold=0.001, one step, no warm-up")
            p = tr["prof"]
            shift = int(os.environ.get("FGOICP_TRIM_SAMPLE", "5"))
            sel_rows, sel_fallbacks, sel_members = tr["solver"].registration.trim_stats()
            one_pass = shift > 0 and sel_rows > 0
            frac_fb = sel_fallbacks / sel_rows if sel_rows else 0.0
            samp = 1.0 / (1 << shift) if shift > 0 else 0.0
            extra = {"select_kernel": "trim_rows_sampled_kernel" if one_pass else "trim_rows_kernel", "select_kernel_ms": p["select_ms"], "bounds_kernel_ms": p["kernel_ms"],
                     # one pass over the row + its 1/2^shift sample (+ two more passes for a row whose bracket failed its exact check) | two passes
                     "select_bytes_per_row": (4.0 * tr["ns"] * (1.0 + samp + 2.0 * frac_fb)) if one_pass else 2 * 4.0 * tr["ns"],
                     "bounds_write_bytes_per_row": 4.0 * tr["ns"] * (1.0 + (samp if one_pass else 0.0)),
                     "select_rows": sel_rows, "select_rows_done_again_in_two_passes": sel_fallbacks,
                     "select_bracket_members_per_row": (sel_members / max(1, sel_rows - sel_fallbacks)) if one_pass else None,
                     "bnb_without_icp_GBps_algorithmic_rank0": tr["stats"]["trans_cubes"] * unit_bytes(tr["ns"]) / (tr["stats"]["seconds_total"] - tr["stats"]["seconds_icp"]) / 1e9,
                     "model": "algorithmic bytes as for the untrimmed operator (SURVEY 8d); on top of them the trimmed path writes 4 B per point-row (e = max(d, 0)) plus a 1/32 "
                              "sample of it, and the selection kernel, which runs NEXT TO the bounds kernel on a side stream, reads the sample and then the row ONCE "
                              "(bracket from the sample, verified exactly; round 2 read every row twice)" if one_pass else
                              "algorithmic bytes as for the untrimmed operator (SURVEY 8d); on top of them the trimmed path writes 4 B per point-row (e = max(d, 0)) and the "
                              "selection kernel, which runs NEXT TO the bounds kernel on a side stream, reads each row twice"}
            extra["limited_by"] = "closest of the three to the HBM roof: measured traffic (LUT lines + 4 B written per point-row) at ~0.7 of the peak, ~0.9 of the measured copy rate"
            s["roofline"] = roofline(tr, pmc_all.get("trimmed"), extra)
            utilisation(s["roofline"], pmc_extra.get("trimmed"))
            s["icp_latency"] = icp_latency(tr)
            line["trimmed_1m_outliers"] = s
            tr["solver"].close()
        guarded("trimmed", _leg_trimmed)

    if rank == 0:
        if world == 1 and want("cpu_baseline") and not a.no_cpu_baseline and a.only in (None, "cpu_baseline"):
            gleg = dflt
            if gleg is None:  # --only cpu_baseline: the GPU run it is compared with
                gleg = run_leg(env, fg, tgt, src, a.lut_resolution, 1e-3, sched, K, 1, 1)
            line["cpu_baseline"] = cpu_baseline(fg, gleg["reg"], tgt, src, a.lut_resolution, 1e-3, 1 if a.schedule == "round" else 0, K, a.cpu_seconds, gleg)
        if a.only and a.only != "headline":
            line["only"] = a.only
        print(json.dumps(line), flush=True)
    for leg in (head, dflt):
        if leg is not None:
            leg["solver"].close()
    if env.ex is not None and hasattr(env.ex, "close"):
        env.ex.close()  # the RCCL communicator goes before the process group and well before interpreter teardown
    if env.dist is not None:
        env.dist.barrier()
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
