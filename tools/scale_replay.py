"""Strong-scaling estimate on ONE GPU, all inside libfgoicp_amd.so (fgoicp_multi_*).  W ranks run together on device 0 over the
in-process transport while every rank records what each exchange returned; then every rank's share is replayed ALONE on the GPU
against its recording, which times what that rank would do on an MI355X of its own — everything but the latency of the two small
RCCL collectives per round.

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
    out = {"workload": workload, "world": world, "mse_threshold": mse, "T1_s": t1, "subcubes_1": int(st1["trans_cubes"]), "rounds_1": int(st1["rounds"]),
           "T_rank_s": times, "subcubes_rank": subs, "seconds_icp_rank": icps, "rounds": rounds,
           "estimated_speedup": t1 / max(times), "estimated_efficiency": t1 / max(times) / world,
           "ideal_if_balanced_speedup": t1 / (sum(times) / world), "same_optimum": bool(abs(float(m.get_best_error()) - e1) <= 1e-5 * e1),
           "all_ranks_together_on_one_gpu_s": together,
           "note": "each rank's share replayed alone on one GPU against the recorded exchange results (fgoicp_multi_replay_rank); the latency of the "
                   "2 small RCCL collectives per round is not included"}
    print(json.dumps(out), flush=True)
    m.close()


if __name__ == "__main__":
    main()
