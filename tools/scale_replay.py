"""Strong-scaling estimate on ONE GPU, all inside libfgoicp_amd.so (fgoicp_multi_*).  W ranks run together on device 0 over the
in-process transport while every rank records what each exchange returned; then every rank's share is replayed ALONE on the GPU
against its recording, which times what that rank would do on an MI355X of its own — everything but the latency of the two small
RCCL collectives per round, which is measured separately: the same two collectives (all-reduce MIN of one float, all-gather of the
round's payload) through fgoicp_rccl_* with a communicator of ONE rank — staging copy, RCCL kernel, copy back, stream wait: the
software path every rank pays per round; the xGMI hops of a real 8-rank ring add single-digit microseconds per step on top
(MI355X_MICROARCH.md) and are not measured here.  `estimated_speedup_with_collectives` charges it once per exchange.
FGOICP_LATE_ICP (0 / 1) selects whether a round's triggered ICP runs overlap the next round (driver.hpp).
FGOICP_COOP_ICP (default 1) = cooperative refinements: every ICP run is executed by all ranks together (1 / world of the source per
scan, two device all-gathers of 4 B per source point per iteration).  A replayed rank computes its own chunk and uploads the others'
from the recording (a pageable host-to-device copy of (W-1)/W of the buffer stands in for the transfer); on top of that the software
path of an in-place RCCL all-gather on device memory (one-rank communicator, same buffer size) is measured and charged twice per ICP
iteration, and the xGMI payload time is MODELLED (not measurable on one GPU): (W-1)/W of the buffer at 50 GB/s, a third of one link.

FGOICP_REPLAY_TRIM=0.2 (with the workload synthetic1m_outliers, mse 1e-3): the trimmed run of BASELINE configs[4].
FGOICP_REPLAY_SCHEDULE=serial: the reference's exact order on both sides (one GPU: SERIAL; W ranks: SERIAL, evaluations sharded).

    python tools/scale_replay.py <world> [workload] [mse] [res] [repeats]     -> one JSON line
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def note(msg):
    """progress on stderr (the runner keeps it in a file under gpurun_out/: a long, silent run is taken for hung on the GPU box)"""
    print(f"[scale_replay {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    import fgoicp_amd as fg
    world = int(sys.argv[1]); workload = sys.argv[2] if len(sys.argv) > 2 else "bunny"
    mse = float(sys.argv[3]) if len(sys.argv) > 3 else 5e-5; res = float(sys.argv[4]) if len(sys.argv) > 4 else 0.005
    repeats = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    tgt, src, _, _ = fg.synth.workload(workload, angle_deg=150.0, min_angle_deg=110.0)
    serial = os.environ.get("FGOICP_REPLAY_SCHEDULE", "round") == "serial"
    sched = fg.SCHEDULE_SERIAL if serial else fg.SCHEDULE_ROUND
    trim = float(os.environ.get("FGOICP_REPLAY_TRIM", "0"))  # e.g. 0.2 with the workload synthetic1m_outliers (BASELINE configs[4])
    note(f"workload {workload} generated; creating the one-GPU solver")
    one = fg.FastGoICP(tgt, src, res, mse, schedule=sched, round_width=1 if serial else 0, device=0, trim_fraction=trim)
    t1 = 1e30
    for _ in range(repeats + 1):
        t0 = time.perf_counter(); R1, _t = one.run(); t1 = min(t1, time.perf_counter() - t0)
    st1 = one.stats()
    e1 = float(one.get_best_error())
    one.close()
    note(f"one GPU: {t1:.3f} s; creating {world} ranks on device 0")
    m = fg.MultiGoICP(tgt, src, res, mse, devices=[0] * world, transport=fg.TRANSPORT_IN_PROCESS, schedule=sched, round_width=1 if serial else 0, trim_fraction=trim)
    m.set_record(True)
    note("ranks created; recorded run of all ranks together")
    t0 = time.perf_counter(); R, t = m.run(); together = time.perf_counter() - t0
    note(f"together: {together:.3f} s; replaying every rank alone")
    times, subs, icps, rounds, iters_rank = [], [], [], None, []
    for r in range(world):
        best = 1e30
        for _ in range(repeats + 1):  # the first pass warms up
            best = min(best, m.replay_rank(r))
        st = m.stats(r)
        note(f"rank {r} alone: {best:.3f} s")
        times.append(best); subs.append(int(st["trans_cubes"])); icps.append(float(st["seconds_icp"])); rounds = int(st["rounds"]); iters_rank.append(int(st.get("icp_iters", 0)))
    host_ex = [m.recorded(r)[0] for r in range(world)]
    dev_gathers = m.recorded(0)[1]
    # the collectives on the RCCL software path (world 1: all a one-GPU box can form)
    ex = fg.RcclExchange(0, 1, fg.rccl_unique_id(), 0)
    import ctypes as C
    per = 13 + 2 * 64
    a1 = (C.c_float * 1)(1.0); snd = (C.c_float * per)(); rcv = (C.c_float * per)()

    def timed(fn, n=200):
        for _ in range(20):
            fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n
    t_ar = timed(lambda: ex.struct.allreduce_min(a1, 1, ex.struct.user))
    t_ag = timed(lambda: ex.struct.allgather(snd, rcv, per, ex.struct.user))
    late = os.environ.get("FGOICP_LATE_ICP", "0")
    coop = os.environ.get("FGOICP_COOP_ICP", "1") != "0"
    # cooperative ICP: the in-place device all-gather, software path (one-rank communicator), on a buffer of the run's size
    t_dev = 0.0
    per_bytes = 4 * (((len(src) + world - 1) // world + 255) & ~255)
    if dev_gathers:
        import torch
        buf = torch.zeros(per_bytes * world, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        t_dev = timed(lambda: ex.struct.allgather_device(buf.data_ptr(), per_bytes * world, ex.struct.user))  # world 1: the whole buffer is this rank's chunk
    note("closing the one-rank RCCL communicator")
    ex.close()
    note("communicator closed")
    band = 1e-5 if (late == "0" or coop) else 2e-3
    t_wire = per_bytes * (world - 1) / 50e9
    t_host = max(t_ar, t_ag)  # every recorded host-side collective is charged the dearer of the two
    with_coll = [times[r] + host_ex[r] * t_host + dev_gathers * (t_dev + t_wire) for r in range(world)]
    out = {"workload": workload, "trim_fraction": trim, "schedule": "serial" if serial else "round", "world": world, "mse_threshold": mse, "late_icp": late, "coop_icp": coop, "T1_s": t1, "subcubes_1": int(st1["trans_cubes"]), "rounds_1": int(st1["rounds"]),
           "T_rank_s": times, "subcubes_rank": subs, "seconds_icp_rank": icps, "rounds": rounds,
           "estimated_speedup": t1 / max(times), "estimated_efficiency": t1 / max(times) / world,
           "allreduce_us_rccl_world1": t_ar * 1e6, "allgather_us_rccl_world1": t_ag * 1e6, "host_exchanges_rank": host_ex,
           "estimated_speedup_with_collectives": t1 / max(with_coll),
           "icp_iterations_rank": iters_rank, "device_allgathers": dev_gathers, "device_allgather_bytes_per_rank": per_bytes,
           "device_allgather_us_rccl_world1": t_dev * 1e6, "device_allgather_wire_us_modelled_50GBps": t_wire * 1e6, "seconds_icp_1": float(st1["seconds_icp"]),
           "ideal_if_balanced_speedup": t1 / (sum(times) / world), "same_optimum": bool(abs(float(m.get_best_error()) - e1) <= band * e1), "same_optimum_band": band,
           "best_sse": float(m.get_best_error()), "best_sse_1": e1,
           "all_ranks_together_on_one_gpu_s": together,
           "note": "each rank's share replayed alone on one GPU against the recorded exchange results (fgoicp_multi_replay_rank); "
                   "estimated_speedup excludes the collectives, estimated_speedup_with_collectives adds the measured RCCL software-path latency per exchange"}
    print(json.dumps(out), flush=True)
    note("closing the multi-rank object")
    m.close()
    note("multi-rank object closed; leaving main()")


if __name__ == "__main__":
    main()
    note("main() returned; interpreter exit follows")
