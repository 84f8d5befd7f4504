#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>.txt (tools/pmc_kernel.sh) -> per-launch figures: clock, MFMA pipe utilisation, wave-cycle split,
HBM bytes (FETCH_SIZE x2 for 16-B/lane streaming reads as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as is; both
are reported in KiB).  usage: pmc_summary.py <tag> [<tag> ...]  -> JSON on stdout."""
import json
import re
import sys


def load(tag):
    vals, durs = {}, []
    for line in open("gpurun_out/pmc_%s.txt" % tag):
        m = re.match(r"PMC (\S+)\s+per-launch ([0-9.e+]+)(?: us)?\s+\(n=(\d+)", line)
        if not m:
            continue
        if m.group(1).startswith("duration["):
            durs.append(float(m.group(2)))
        else:
            vals[m.group(1)] = float(m.group(2))
            vals["_n"] = int(m.group(3))
    return vals, durs


def summary(tag):
    v, durs = load(tag)
    dur = sum(durs) / len(durs)
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the 8 XCDs
    out = {"tag": tag, "launches_profiled": v["_n"], "duration_us_under_profiler": round(dur, 1),
           "clock_ghz": round(cyc / dur / 1e3, 3),
           "mfma_instructions": v.get("SQ_INSTS_MFMA", 0), "valu_instructions": v.get("SQ_INSTS_VALU", 0),
           "mfma_pipe_utilisation": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), 4),
           "wave_cycles_active_frac": round(v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3),
           "wave_cycles_valu_frac": round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 3),
           "wave_cycles_lds_frac": round(v["SQ_ACTIVE_INST_LDS"] / v["SQ_WAVE_CYCLES"], 3),
           "wave_cycles_wait_any_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3),
           "wave_cycles_wait_inst_frac": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3),
           "lds_bank_conflict_cycles": v.get("SQ_LDS_BANK_CONFLICT", 0), "lds_active_cycles": v.get("SQ_LDS_IDX_ACTIVE", 0),
           "fetch_bytes_reported": v["FETCH_SIZE"] * 1024, "fetch_bytes_corrected_x2": v["FETCH_SIZE"] * 2048,
           "write_bytes": v["WRITE_SIZE"] * 1024}
    out["traffic_bytes_per_launch"] = out["fetch_bytes_corrected_x2"] + out["write_bytes"]
    out["hbm_tb_per_s"] = round(out["traffic_bytes_per_launch"] / dur / 1e6, 3)
    return out


if __name__ == "__main__":
    print(json.dumps({t: summary(t) for t in sys.argv[1:]}, indent=1))
