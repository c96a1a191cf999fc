"""Strong-scaling estimate on ONE GPU, all inside libfgoicp_amd.so (fgoicp_multi_*).  W ranks run together on device 0 over the
in-process transport while every rank records what each exchange returned; then every rank's share is replayed ALONE on the GPU
against its recording, which times what that rank would do on an MI355X of its own — everything but the latency of the two small
RCCL collectives per round, which is measured separately: the same two collectives (all-reduce MIN of one float, all-gather of the
round's payload) through fgoicp_rccl_* with a communicator of ONE rank — staging copy, RCCL kernel, copy back, stream wait: the
software path every rank pays per round; the xGMI hops of a real 8-rank ring add single-digit microseconds per step on top
(MI355X_MICROARCH.md) and are not measured here.  `estimated_speedup_with_collectives` charges it once per exchange.
FGOICP_LATE_ICP (0 / 1) selects whether a round's triggered ICP runs overlap the next round (driver.hpp).

    python tools/scale_replay.py <world> [workload] [mse] [res] [repeats]     -> one JSON line
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import fgoicp_amd as fg
    world = int(sys.argv[1]); workload = sys.argv[2] if len(sys.argv) > 2 else "bunny"
    mse = float(sys.argv[3]) if len(sys.argv) > 3 else 5e-5; res = float(sys.argv[4]) if len(sys.argv) > 4 else 0.005
    repeats = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    tgt, src, _, _ = fg.synth.workload(workload, angle_deg=150.0, min_angle_deg=110.0)
    one = fg.FastGoICP(tgt, src, res, mse, schedule=fg.SCHEDULE_ROUND, round_width=0, device=0)
    t1 = 1e30
    for _ in range(repeats + 1):
        t0 = time.perf_counter(); R1, _t = one.run(); t1 = min(t1, time.perf_counter() - t0)
    st1 = one.stats()
    e1 = float(one.get_best_error())
    one.close()
    m = fg.MultiGoICP(tgt, src, res, mse, devices=[0] * world, transport=fg.TRANSPORT_IN_PROCESS)
    m.set_record(True)
    t0 = time.perf_counter(); R, t = m.run(); together = time.perf_counter() - t0
    times, subs, icps, rounds = [], [], [], None
    for r in range(world):
        best = 1e30
        for _ in range(repeats + 1):  # the first pass warms up
            best = min(best, m.replay_rank(r))
        st = m.stats(r)
        times.append(best); subs.append(int(st["trans_cubes"])); icps.append(float(st["seconds_icp"])); rounds = int(st["rounds"])
    # the per-round collectives on the RCCL software path (world 1: all a one-GPU box can form)
    ex = fg.RcclExchange(0, 1, fg.rccl_unique_id(), 0)
    import ctypes as C
    per = 13 + 2 * 64
    a1 = (C.c_float * 1)(1.0); snd = (C.c_float * per)(); rcv = (C.c_float * per)()
    for _ in range(20):
        ex.struct.allreduce_min(a1, 1, ex.struct.user); ex.struct.allgather(snd, rcv, per, ex.struct.user)
    t0 = time.perf_counter()
    for _ in range(200):
        ex.struct.allreduce_min(a1, 1, ex.struct.user); ex.struct.allgather(snd, rcv, per, ex.struct.user)
    t_ex = (time.perf_counter() - t0) / 200
    ex.close()
    exchanges = rounds + (1 if os.environ.get("FGOICP_LATE_ICP", "1") != "0" else 0)
    band = 1e-5 if os.environ.get("FGOICP_LATE_ICP", "1") == "0" else 2e-3
    out = {"workload": workload, "world": world, "mse_threshold": mse, "late_icp": os.environ.get("FGOICP_LATE_ICP", "1"), "T1_s": t1, "subcubes_1": int(st1["trans_cubes"]), "rounds_1": int(st1["rounds"]),
           "T_rank_s": times, "subcubes_rank": subs, "seconds_icp_rank": icps, "rounds": rounds,
           "estimated_speedup": t1 / max(times), "estimated_efficiency": t1 / max(times) / world,
           "exchange_us_rccl_world1": t_ex * 1e6, "exchanges": exchanges,
           "estimated_speedup_with_collectives": t1 / (max(times) + exchanges * t_ex),
           "ideal_if_balanced_speedup": t1 / (sum(times) / world), "same_optimum": bool(abs(float(m.get_best_error()) - e1) <= band * e1), "same_optimum_band": band,
           "best_sse": float(m.get_best_error()), "best_sse_1": e1,
           "all_ranks_together_on_one_gpu_s": together,
           "note": "each rank's share replayed alone on one GPU against the recorded exchange results (fgoicp_multi_replay_rank); "
                   "estimated_speedup excludes the collectives, estimated_speedup_with_collectives adds the measured RCCL software-path latency per exchange"}
    print(json.dumps(out), flush=True)
    m.close()


if __name__ == "__main__":
    main()
