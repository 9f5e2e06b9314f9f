#!/usr/bin/env python3
"""Condense tools/profile_r03.sh output into small files fit for profiles/:
     r03_<workload>_kernel_stats.csv   rows of rocprofv3's --stats summary for this library's kernels (verbatim columns)
     r03_<workload>_counters.json      per-launch averages of the PMC passes for the dominant kernel + derived figures
   usage: summarize_r02.py <raw dir> <out dir>     (then copy <out dir>/* into profiles/)
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE (KiB) x 2 on gfx950, WRITE_SIZE (KiB) as read."""
import collections
import csv
import glob
import json
import os
import sys

raw, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
OURS = ("k_trace_", "k_gen_", "k_mon_", "k_stream_")
RECORD_BYTES = {"cfg2": 104, "cfg4": 104, "cfg3": 56, "cfg5": 56}  # per ray record and per segment record


def algorithmic_bytes(wdir, w):
    """(rays + segments) x record bytes per launch, from what the profiled program printed (tools/profile_workload.py)."""
    import re
    if w not in RECORD_BYTES:
        return None
    try:
        text = open(os.path.join(wdir, "trace.log")).read()
    except OSError:
        return None
    m = re.search(r"(\d+) (?:rays|ray-wavelength pairs), (\d+) segments per trace", text)
    return (int(m.group(1)) + int(m.group(2))) * RECORD_BYTES[w] if m else None
for wdir in sorted(glob.glob(os.path.join(raw, "*"))):
    w = os.path.basename(wdir)
    stats = glob.glob(os.path.join(wdir, "trace", "*", "*_kernel_stats.csv"))
    if not stats:
        continue
    rows = [r for r in csv.DictReader(open(stats[0])) if any(k in r["Name"] for k in OURS)]
    if not rows:
        continue
    with open(os.path.join(dst, f"r03_{w}_kernel_stats.csv"), "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(rows)
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    short = top["Name"].split("(")[0].replace("void ", "")
    rec = {"workload": w, "kernel": short, "calls": int(top["Calls"]), "avg_ns": float(top["AverageNs"]),
           "min_ns": float(top["MinNs"]), "max_ns": float(top["MaxNs"]),
           "all_kernels_ns_per_run": {r["Name"].split("(")[0].replace("void ", ""): float(r["TotalDurationNs"]) for r in rows}}
    acc = collections.defaultdict(list)
    for part in ("sq", "fetch", "write"):
        for f in glob.glob(os.path.join(wdir, part, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].split("(")[0].replace("void ", "") == short:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    rec.setdefault("dispatch", {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")})
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    rec["counters_per_launch"] = c
    d = {}
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_THREAD_CYCLES_VALU" in c and c["SQ_ACTIVE_INST_VALU"]:
        d["active_lane_fraction"] = c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
        d["wait_share_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
        # SQ_ACTIVE_INST_* count quad-cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        d["valu_issue_fraction"] = 4 * c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_INSTS_VALU" in c and c.get("SQ_WAVES"):
        d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_read_bytes_fetch_x2"] = c["FETCH_SIZE"] * 1024 * 2
        d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        d["hbm_bytes"] = d["hbm_read_bytes_fetch_x2"] + d["hbm_write_bytes"]
        d["hbm_gbs"] = d["hbm_bytes"] / rec["avg_ns"]
        alg = algorithmic_bytes(wdir, w)
        if alg:
            d["algorithmic_bytes"] = alg
            d["traffic_over_algorithmic"] = d["hbm_bytes"] / alg
            d["hbm_frac_algorithmic"] = alg / rec["avg_ns"] / 8000.0
    rec["derived"] = d
    json.dump(rec, open(os.path.join(dst, f"r03_{w}_counters.json"), "w"), indent=1)
    print(w, short, f"{rec['avg_ns'] / 1e6:.3f} ms", {k: round(v, 4) for k, v in d.items() if k not in ("hbm_read_bytes_fetch_x2", "hbm_write_bytes", "hbm_bytes")})
