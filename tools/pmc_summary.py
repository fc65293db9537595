#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/profile_round.sh.

    python tools/pmc_summary.py gpurun_out/<tag> [kernel-name-substring]

Prints one JSON object: per-launch FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports
them) and the SQ/GRBM counters of the dominant kernel (default: k_align), plus the
derived VALU-issue occupancy.  bench.py reads the committed copy under profiles/.
"""
import csv
import glob
import json
import os
import sys


def counter_rows(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    return rows


def per_launch(rows, kernel, counter):
    by_dispatch = {}
    for r in rows:
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            by_dispatch[r["Dispatch_Id"]] = by_dispatch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return [by_dispatch[k] for k in sorted(by_dispatch, key=int)]


def kernel_stats(d, kernel):
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if kernel in r["Name"]:
                    return {"name": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                            "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}
    return None


def main():
    out_dir = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else "k_align"
    res = {"kernel_substring": kernel}
    res["kernel_stats"] = kernel_stats(os.path.join(out_dir, "stats"), kernel)
    fetch = per_launch(counter_rows(os.path.join(out_dir, "pmc_fetch")), kernel, "FETCH_SIZE")
    write = per_launch(counter_rows(os.path.join(out_dir, "pmc_write")), kernel, "WRITE_SIZE")
    res["FETCH_SIZE_KB_per_launch"] = fetch
    res["WRITE_SIZE_KB_per_launch"] = write
    sq_rows = counter_rows(os.path.join(out_dir, "pmc_sq"))
    sq = {}
    for c in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
              "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        v = per_launch(sq_rows, kernel, c)
        if v:
            sq[c] = sum(v) / len(v)
    res["valu_pass"] = sq
    if res["kernel_stats"] and "GRBM_GUI_ACTIVE" in sq and "SQ_ACTIVE_INST_VALU" in sq:
        ms = res["kernel_stats"]["avg_ms"]
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_ACTIVE_INST_* count quad-cycles
        clk = sq["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e-3) / 1e9
        res["derived"] = {
            "effective_clock_GHz": clk,
            "valu_busy_frac": 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (1024.0 * sq["GRBM_GUI_ACTIVE"] / 8.0),
            "valu_wave_instr_per_launch": sq.get("SQ_INSTS_VALU"),
            "note": "SQ_ACTIVE_INST_VALU counts quad-cycles; busy fraction = 4*SQ_ACTIVE_INST_VALU / "
                    "(1024 SIMDs * GRBM_GUI_ACTIVE/8)",
        }
    res["note"] = ("FETCH_SIZE on gfx950 reports half the bytes of wide (16 B/lane) coalesced reads "
                   "(MI355X_MICROARCH.md, HBM); narrower accesses are uncalibrated, value taken at face")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
