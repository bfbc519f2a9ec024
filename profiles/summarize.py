#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (profiles/run_profile.sh) into the small files
committed under profiles/: <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and
<tag>_pmc_summary.json (per-kernel SQ / FETCH_SIZE / WRITE_SIZE aggregates, per launch)."""
import collections
import csv
import json
import os
import shutil
import sys


def agg(path, key_len=120):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:key_len]
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]:
            out[k]["_duration_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        n[k].add(r["Dispatch_Id"])
    return out, {k: len(v) for k, v in n.items()}


def main(tag, suffix=""):
    """suffix: appended to the output names, e.g. "_bf16x3" (bench.py's fallback looks for r*_pmc_summary_<precision>.json)"""
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(os.path.dirname(here), "gpurun_out", "prof_" + tag)
    shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(here, tag + "_kernel_stats" + suffix + ".csv"))
    summ = {}
    sq, nsq = agg(os.path.join(src, "pmc_sq", "p_counter_collection.csv"))
    fe, nfe = agg(os.path.join(src, "pmc_fetch", "p_counter_collection.csv"))
    wr, nwr = agg(os.path.join(src, "pmc_write", "p_counter_collection.csv"))
    for k, v in sq.items():
        if not ("lrp::" in k):
            continue
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        d = {"launches": nsq[k]}
        if gui:
            # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs each able to hold one MFMA-cycle per cycle
            d["mfma_busy_frac"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), 4)
            if v.get("_duration_ns"):
                # shader clock the kernel actually ran at: busy cycles of one XCD / wall time of its dispatches
                d["shader_clock_ghz"] = round(gui / 8 / v["_duration_ns"], 3)
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            d["wait_any_frac"] = round(v.get("SQ_WAIT_ANY", 0.0) / wc, 4)
            d["wait_inst_frac"] = round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
            d["active_frac"] = round(v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)
            d["lds_bank_conflict_per_wave_cycle"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / wc, 5)
        if k in fe:
            # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half of a 16 B/lane coalesced read stream (x2)
            d["fetch_bytes_per_launch_x2corr"] = round(2 * fe[k]["FETCH_SIZE"] * 1024 / nfe[k])
        if k in wr:
            d["write_bytes_per_launch"] = round(wr[k]["WRITE_SIZE"] * 1024 / nwr[k])
        summ[k] = d
    json.dump(summ, open(os.path.join(here, tag + "_pmc_summary" + suffix + ".json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(summ, indent=1, sort_keys=True)[:3000])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03", sys.argv[2] if len(sys.argv) > 2 else "")
