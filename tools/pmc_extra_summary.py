"""Condenses the SQ / TA / TCP passes of tools/pmc_extra.sh (profiles/<tag>_<leg>_pmc_extra*.json, per-kernel counter sums) into
profiles/bench_pmc_extra.json: per leg the bounds kernel's per-launch instruction and cache figures that bench.py turns into the
`valu` / `l1` / `ta` utilisations of its roofline objects.

    python tools/pmc_extra_summary.py <tag> <leg>=<file.json>[,<file.json>...] [<leg>=...]

Derived figures (cycle-based ones come from the profiled pass itself and do not depend on the clock):
  cycles_per_launch          GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back")
  valu_issue_utilisation     SQ_INSTS_VALU x 2 cycles (a wave64 VALU instruction issues over 2 cycles on a SIMD-32) / (1024 SIMDs x cycles)
  l1_hit_rate                1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES
  l1_miss_latency_cycles     TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ
  l1_pending_stall_frac      TCP_PENDING_STALL_CYCLES / (256 TCPs x cycles)
  ta_busy_frac               TA_BUSY_avr / cycles           (average over the texture addressers)
  ta_addr_stalled_frac       TA_ADDR_STALLED_BY_TC_CYCLES / (256 x cycles)
  l1_accesses_per_clock_cu   TCP_TOTAL_CACHE_ACCESSES / (256 TCPs x cycles): cache-line accesses the L1 of a CU handles per clock
  wave_wait_frac             SQ_WAIT_ANY / SQ_WAVE_CYCLES   (share of wave lifetime parked in s_waitcnt / barriers)
  wave_issue_stall_frac      SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    out = {}
    for spec in sys.argv[2:]:
        leg, files = spec.split("=")
        c = {}
        for f in files.split(","):
            d = json.load(open(f))
            for k, v in (d.get("bounds_item_kernel") or d.get("bounds_sorted_kernel") or {}).items():
                c[k] = v["per_dispatch"]
        if not c:
            continue
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        e = {"kernel": "bounds_item_kernel", "source": [os.path.relpath(f, REPO) for f in files.split(",")], "cycles_per_launch": cyc, "counters_per_launch": c}
        g = c.get
        if g("SQ_INSTS_VALU") and cyc:
            e["valu_insts_per_launch"] = g("SQ_INSTS_VALU")
            e["valu_issue_utilisation"] = g("SQ_INSTS_VALU") * 2.0 / (1024.0 * cyc)
        if g("SQ_INSTS_SALU"):
            e["salu_insts_per_launch"] = g("SQ_INSTS_SALU")
        if g("SQ_INSTS_VMEM_RD"):
            e["vmem_rd_insts_per_launch"] = g("SQ_INSTS_VMEM_RD")
        if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
            e["l1_accesses_per_launch"] = g("TCP_TOTAL_CACHE_ACCESSES_sum")
            e["l1_hit_rate"] = 1.0 - g("TCP_TCC_READ_REQ_sum", 0.0) / g("TCP_TOTAL_CACHE_ACCESSES_sum")
            if cyc:
                e["l1_accesses_per_clock_cu"] = g("TCP_TOTAL_CACHE_ACCESSES_sum") / (256.0 * cyc)
        if g("TCP_TCC_READ_REQ_sum"):
            e["l1_miss_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / g("TCP_TCC_READ_REQ_sum")
        if g("TCP_PENDING_STALL_CYCLES_sum") and cyc:
            e["l1_pending_stall_frac"] = g("TCP_PENDING_STALL_CYCLES_sum") / (256.0 * cyc)
        if g("TA_BUSY_avr") and cyc:
            e["ta_busy_frac"] = g("TA_BUSY_avr") / cyc
        if g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") and cyc:
            e["ta_addr_stalled_frac"] = g("TA_ADDR_STALLED_BY_TC_CYCLES_sum") / (256.0 * cyc)
        if g("TA_DATA_STALLED_BY_TC_CYCLES_sum") and cyc:
            e["ta_data_stalled_frac"] = g("TA_DATA_STALLED_BY_TC_CYCLES_sum") / (256.0 * cyc)
        if g("SQ_WAVE_CYCLES"):
            if g("SQ_WAIT_ANY"):
                e["wave_wait_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
            if g("SQ_WAIT_INST_ANY"):
                e["wave_issue_stall_frac"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        out[leg] = e
    path = os.path.join(REPO, "profiles", "bench_pmc_extra.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    for leg, e in out.items():
        print(leg, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if k not in ("counters_per_launch", "source")})


if __name__ == "__main__":
    main()
