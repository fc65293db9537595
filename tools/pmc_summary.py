#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/profile_round.sh for one kernel.

    python tools/pmc_summary.py gpurun_out/<tag> [kernel-name-substring]

Prints one JSON object: per-launch FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports them), the
SQ/GRBM counters of the kernel (averaged over its launches), derived figures, and `source_sha`, the
hash of the kernel's source file at profiling time -- bench.py quotes the committed copy under
profiles/ only while the source still hashes to it.
"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = {"k_align": ["align.hip"], "k_msa_pairwise": ["msa_pairwise.hip", "msa_common.hpp"], "k_consensus_code": ["consensus.hip", "msa_common.hpp"],
           "k_m2_group": ["msa2.hip", "msa_common.hpp"], "k_align_wide": ["align.hip"]}
# share of fp64 instructions (4 issue cycles per wave64 instruction on a SIMD-32; everything else
# 2) in the kernel's VALU stream, from the disassembly of its main loop
FP64_SHARE = {"k_align": 0.75, "k_msa_pairwise": 0.0, "k_consensus_code": 0.18, "k_m2_group": 0.0, "k_align_wide": 0.53}


def source_sha(names):
    h = hashlib.sha256()
    for nm in names:
        with open(os.path.join(ROOT, "sarlacc_amd", "csrc", nm), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


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
    best = None
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if ((kernel + "<") in r["Name"]) if kernel == "k_align" else (kernel in r["Name"]):   # ("k_align" is also the start of k_align_wide_q)
                    rec = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                           "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6,
                           "total_ms": float(r["TotalDurationNs"]) / 1e6}
                    if best is None or rec["total_ms"] > best["total_ms"]:
                        best = rec
    return best


def main():
    out_dir = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else "k_align"
    res = {"kernel_substring": kernel, "source_sha": source_sha(SOURCES.get(kernel, []))}
    bj = os.path.join(out_dir, "bench.json")
    if os.path.exists(bj):
        try:
            with open(bj) as fh:
                b = json.loads(fh.read().strip().splitlines()[-1])
            res["workload"] = (b["config"]["workload"] if kernel == "k_align" else b.get("quality_align_2kb", {}).get("note") if kernel == "k_align_wide"
                               else b.get("pipeline", {}).get("workload"))
            # what the SAME box measured without the profiler (tools/profile_round.sh runs the plain bench first): a reader of a
            # bench line that quotes these counters sees at once whether its own timings come from another machine
            res["same_box_bench"] = ({"kernel_ms": b.get("kernel_ms"), "value": b.get("value"), "ms_per_step": b.get("ms_per_step")} if kernel == "k_align"
                                     else {"kernel_ms": b.get("quality_align_2kb", {}).get("kernel_ms"), "kernel_gcups": b.get("quality_align_2kb", {}).get("kernel_gcups")} if kernel == "k_align_wide"
                                     else {"kernel_ms": b.get("pipeline", {}).get("kernel_ms"), "pipeline_seconds": b.get("pipeline", {}).get("seconds")})
        except (ValueError, KeyError, IndexError):
            pass
    bx = os.path.join(out_dir, "box.json")
    if os.path.exists(bx):
        try:
            with open(bx) as fh:
                box = json.load(fh)
            res["box"] = {"hostname": box.get("hostname"), "gpu": [a for a in box.get("gpu_agents", []) if "Uuid" in a and "GPU" in a][:1]}
        except ValueError:
            pass
    res["kernel_stats"] = kernel_stats(os.path.join(out_dir, "stats"), kernel)
    # the profiler's average duration beside what the same box measured without it (HIP events inside the plain bench run
    # that tools/profile_round.sh makes first): the spread between the two is stated where the fractions are computed
    unprof = (res.get("same_box_bench") or {}).get("kernel_ms")
    if res["kernel_stats"] and isinstance(unprof, (int, float)) and unprof > 0:
        res["kernel_stats"]["same_box_unprofiled_ms"] = unprof
        res["kernel_stats"]["profiled_over_unprofiled"] = res["kernel_stats"]["avg_ms"] / unprof
    name = res["kernel_stats"]["name"] if res["kernel_stats"] else kernel
    # counters of the dominant instantiation only (band classes / template variants are separate kernels)
    key = name.split("(")[0] if res["kernel_stats"] else kernel
    res["FETCH_SIZE_KB_per_launch"] = per_launch(counter_rows(os.path.join(out_dir, "pmc_fetch")), key, "FETCH_SIZE")
    res["WRITE_SIZE_KB_per_launch"] = per_launch(counter_rows(os.path.join(out_dir, "pmc_write")), key, "WRITE_SIZE")
    # HBM bytes per launch after the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-byte
    # requests as 64 bytes, so reads are doubled; WRITE_SIZE is taken as reported
    fe, wr = res["FETCH_SIZE_KB_per_launch"], res["WRITE_SIZE_KB_per_launch"]
    if fe and wr:
        # the full-size launches only (a bench run also launches the kernel on remainders: groups redone with wider profiles ...)
        big_f = [x for x in fe if x >= 0.5 * max(fe)]
        big_w = [x for x in wr if x >= 0.5 * max(wr)]
        res["traffic_bytes_per_launch"] = (2.0 * sum(big_f) / len(big_f) + sum(big_w) / len(big_w)) * 1024.0
        res["traffic_correction"] = ("2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half the bytes read), mean over the "
                                     "launches within a factor 2 of the largest (%d of %d)" % (len(big_f), len(fe)))
    if kernel == "k_msa_pairwise" and "k_msa_pairwise_bv" in name:
        # The pairwise stage on bit vectors is many launches: fill and walk kernels per chunk of batches plus one expansion.
        # Its traffic is the sum over all of them, per stage (= per launch of k_msa_moves_expand); the counters below stay those
        # of the dominant kernel (the fill).
        stage = ("k_msa_pairwise_bv", "k_msa_moves_expand")
        def stages(sub, counter):
            # per stage: the counter summed over the stage's launches; a stage ends with its k_msa_moves_expand
            per = {}
            for r in counter_rows(os.path.join(out_dir, sub)):
                if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in stage):
                    d = int(r["Dispatch_Id"])
                    per.setdefault(d, [0.0, "k_msa_moves_expand" in r["Kernel_Name"]])[0] += float(r["Counter_Value"])
            out, acc = [], 0.0
            for d in sorted(per):
                acc += per[d][0]
                if per[d][1]:
                    out.append(acc)
                    acc = 0.0
            return out
        sf, sw = stages("pmc_fetch", "FETCH_SIZE"), stages("pmc_write", "WRITE_SIZE")
        if sf and sw:
            # the full-size stages only (the bench also runs the stage on spec v1's shorter job list)
            bf = [x for x in sf if x >= 0.5 * max(sf)]
            bw = [x for x in sw if x >= 0.5 * max(sw)]
            res["stage_launches_profiled"] = len(sf)
            res["fetch_bytes_per_stage"] = 2.0 * sum(bf) / len(bf) * 1024.0
            res["write_bytes_per_stage"] = sum(bw) / len(bw) * 1024.0
            res["traffic_bytes_per_launch"] = res["fetch_bytes_per_stage"] + res["write_bytes_per_stage"]
            res["traffic_scope"] = ("one pairwise stage = all launches of k_msa_pairwise_bv<NW, 1, NS> (fill), <NW, 2, NS> (walk), the second run of the "
                                    "pairs that left their partial records, and k_msa_moves_expand of one call; mean over the %d full-size stages "
                                    "of the bench run (of %d: spec v1's shorter job list is left out)" % (len(bf), len(sf)))
            res["traffic_correction"] = "2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half the bytes read), summed over the stage's launches"
    if kernel == "k_m2_group":
        # The merge stage of spec v2 is up to four instantiations of k_m2_group side by side (1, 4, 8 wavefronts per group) plus
        # k_m2_tree / k_m2_init / k_m2_first / k_m2_extend*: its traffic is the sum over all of them, per stage (a stage starts with its k_m2_tree).
        def stages(sub, counter):
            per = {}
            for r in counter_rows(os.path.join(out_dir, sub)):
                if r["Counter_Name"] == counter and "k_m2_" in r["Kernel_Name"] and "k_m2_jobs" not in r["Kernel_Name"] and "k_m2_write" not in r["Kernel_Name"]:
                    d = int(r["Dispatch_Id"])
                    per.setdefault(d, [0.0, "k_m2_tree" in r["Kernel_Name"]])[0] += float(r["Counter_Value"])
            out = []
            for d in sorted(per):
                if per[d][1] or not out:
                    out.append(0.0)
                out[-1] += per[d][0]
            return out
        sf, sw = stages("pmc_fetch", "FETCH_SIZE"), stages("pmc_write", "WRITE_SIZE")
        if sf and sw:
            bf = [x for x in sf if x >= 0.75 * max(sf)]   # the pipeline's clusters (the pure groups of configs[3] move about half as much)
            bw = [x for x in sw if x >= 0.75 * max(sw)]
            res["stage_launches_profiled"] = len(sf)
            res["fetch_bytes_per_stage"] = 2.0 * sum(bf) / len(bf) * 1024.0
            res["write_bytes_per_stage"] = sum(bw) / len(bw) * 1024.0
            res["traffic_bytes_per_launch"] = res["fetch_bytes_per_stage"] + res["write_bytes_per_stage"]
            res["traffic_bytes_per_stage_all"] = [(2.0 * a + b) * 1024.0 for a, b in zip(sf, sw)]
            res["traffic_scope"] = ("one merge stage = k_m2_tree + k_m2_init + k_m2_first + the launches of k_m2_extend / k_m2_extend_unit + every instantiation of k_m2_group of one call; mean over "
                                    "the %d stages on the pipeline's clusters (of %d: the pure groups of configs[3] are listed in traffic_bytes_per_stage_all)"
                                    % (len(bf), len(sf)))
            res["traffic_correction"] = "2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half the bytes read), summed over the stage's launches"
    sq = {}
    for sub, names in (("pmc_sq", ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
                                   "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")),
                       ("pmc_sq2", ("SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
                                    "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"))):
        rows = counter_rows(os.path.join(out_dir, sub))
        for c in names:
            v = per_launch(rows, key, c)
            if v:
                sq[c] = sum(v) / len(v)
    res["sq"] = sq
    if res["kernel_stats"] and "GRBM_GUI_ACTIVE" in sq:
        ms = res["kernel_stats"]["avg_ms"]
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        cyc = sq["GRBM_GUI_ACTIVE"] / 8.0
        d = {"effective_clock_GHz": cyc / (ms * 1e-3) / 1e9}
        if "SQ_INSTS_VALU" in sq:
            share = FP64_SHARE.get(kernel, 0.0)
            per_instr = 2.0 + 2.0 * share
            d["valu_wave_instr_per_launch"] = sq["SQ_INSTS_VALU"]
            d["valu_issue_frac"] = sq["SQ_INSTS_VALU"] * per_instr / (1024.0 * cyc)
            d["valu_issue_model"] = ("wave64 VALU instruction = 2 issue cycles on a SIMD-32, fp64 = 4; fp64 share %.2f of the "
                                     "stream; fraction = instructions x cycles / (1024 SIMDs x GRBM_GUI_ACTIVE/8)" % share)
        if "SQ_WAVE_CYCLES" in sq and sq["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if c in sq:
                    d[c.lower() + "_per_wave_cycle"] = sq[c] / sq["SQ_WAVE_CYCLES"]
        if "SQ_LDS_IDX_ACTIVE" in sq and sq.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            d["lds_bank_conflict_frac"] = sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"]
        res["derived"] = d
    res["note"] = ("FETCH_SIZE on gfx950 reports half the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM): "
                   "traffic_bytes_per_launch doubles it; the guide calls narrower accesses uncalibrated")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
