#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of tools/round_profile.sh from gpurun_out/ into profiles/ and
derive the per-launch HBM traffic of the tracking kernel (profiles/pmc_traffic.json, read by
bench.py for roofline.traffic).

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies 128-B fabric requests at 64 B, so the read side is doubled; WRITE_SIZE is
exact.  (The guide calibrates that factor on wide coalesced reads; dword gathers through the
same 128-B L2 lines are assumed to follow it — stated in DESIGN.md.)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "config2_ratio_1080p_64spp_cloud512"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(os.path.join(G, pattern))
    return max(files, key=os.path.getmtime) if files else None


def counters(path, kernel="vr_track_kernel"):
    agg = collections.defaultdict(float)
    launches = set()
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            launches.add(r["Dispatch_Id"])
    return dict(agg), max(len(launches), 1)


ks = newest("prof_kt/*/*kernel_stats.csv")
shutil.copy(ks, os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
with open(os.path.join(G, "bench_default.log")) as f:      # the log holds stderr noise too: keep the JSON line
    line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
with open(os.path.join(P, f"{tag}_bench.json"), "w") as f:
    f.write(line + "\n")
out = {}
for name in ("prof_fetch", "prof_write", "prof_sq"):
    f = newest(f"{name}/*/*counter_collection.csv")
    if f:
        c, n = counters(f)
        out[name] = {"launches": n, "counters_sum_over_launches": c}
fetch = out["prof_fetch"]["counters_sum_over_launches"]["FETCH_SIZE"] / out["prof_fetch"]["launches"]
write = out["prof_write"]["counters_sum_over_launches"]["WRITE_SIZE"] / out["prof_write"]["launches"]
hbm = (2.0 * fetch + write) * 1024.0
out["derived"] = {"FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
                  "hbm_bytes_per_launch": hbm, "read_correction": 2.0}
json.dump(out, open(os.path.join(P, f"{tag}_bench_pmc_track_kernel.json"), "w"), indent=1)
tpath = os.path.join(P, "pmc_traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
traffic[workload] = {"hbm_bytes_per_launch": hbm, "source": f"profiles/{tag}_bench_pmc_track_kernel.json"}
if "prof_sq" in out and "SQ_INSTS_VALU" in out["prof_sq"]["counters_sum_over_launches"]:
    sq = out["prof_sq"]
    traffic[workload]["valu_insts_per_launch"] = sq["counters_sum_over_launches"]["SQ_INSTS_VALU"] / sq["launches"]
    traffic[workload]["valu_source"] = f"SQ_INSTS_VALU, profiles/{tag}_bench_pmc_track_kernel.json"
    c = sq["counters_sum_over_launches"]
    traffic[workload]["salu_insts_per_launch"] = c.get("SQ_INSTS_SALU", 0.0) / sq["launches"]
    if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES"):
        traffic[workload]["wave_cycles_waiting"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
        # lanes active per VALU instruction, of 64: what divergence (event phases, the per-trip service blocks) leaves
        traffic[workload]["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] / 64.0
json.dump(traffic, open(tpath, "w"), indent=1)
print(open(os.path.join(P, f"{tag}_bench_kernel_stats.csv")).read()[:600])
print(json.dumps(out["derived"], indent=1))
