#!/usr/bin/env python3
"""gpurun_out/pmc_wl/ (tools/pmc_workloads.sh) -> profiles/<tag>_pmc_workloads.json and the per-workload entries of
profiles/pmc_traffic.json (HBM bytes and VALU instructions per launch of the workload's dominant kernel), which bench.py
reads for roofline.traffic / roofline.valu.  HBM bytes as in tools/collect_profiles.py (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE and WRITE_SIZE in KiB, the read side doubled on gfx950)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
G = os.path.join(ROOT, "gpurun_out", "pmc_wl")
P = os.path.join(ROOT, "profiles")
KERNELS = ("vr_track_kernel", "vr_delta_dda_kernel", "vr_gridmarch_kernel", "vr_march_kernel")


def counters(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                per[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k].add(r["Dispatch_Id"])
    if not per:
        return None, {}, 1
    k = max(per, key=lambda k: sum(per[k].values()))
    return k, dict(per[k]), max(len(launches[k]), 1)


out = {}
tpath = os.path.join(P, "pmc_traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
for f in sorted(glob.glob(os.path.join(G, "*_sq.csv"))):
    key = os.path.basename(f)[:-7]                 # workload_mode
    workload, mode = key.rsplit("_", 1)
    e = {}
    for p in ("fetch", "write", "sq"):
        k, c, n = counters(os.path.join(G, f"{key}_{p}.csv"))
        e[p] = {"kernel": k, "launches": n, "counters_per_launch": {a: b / n for a, b in c.items()}}
    fetch = e["fetch"]["counters_per_launch"]["FETCH_SIZE"]
    write = e["write"]["counters_per_launch"]["WRITE_SIZE"]
    hit, miss = e["write"]["counters_per_launch"].get("TCC_HIT_sum", 0.0), e["write"]["counters_per_launch"].get("TCC_MISS_sum", 0.0)
    bench = json.loads(open(os.path.join(G, f"{key}_bench.json")).read().strip().splitlines()[-1])
    ms = bench["roofline"]["kernel_ms_per_launch"]
    e["derived"] = {"hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0, "read_correction": 2.0,
                    "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
                    "kernel_ms_under_the_counters": ms,
                    "hbm_GBps": (2.0 * fetch + write) * 1024.0 / (ms * 1e-3) / 1e9,
                    "valu_insts_per_launch": e["sq"]["counters_per_launch"].get("SQ_INSTS_VALU"),
                    "loads_bytes_per_launch": bench["roofline"].get("loads", {}).get("bytes_per_launch")}
    out[key] = e
    name = workload if mode == "fast" else f"{workload}:{mode}"
    sq = e["sq"]["counters_per_launch"]
    traffic[name] = {"hbm_bytes_per_launch": e["derived"]["hbm_bytes_per_launch"], "source": f"profiles/{tag}_pmc_workloads.json",
                     "valu_insts_per_launch": e["derived"]["valu_insts_per_launch"],
                     "valu_source": f"SQ_INSTS_VALU, profiles/{tag}_pmc_workloads.json",
                     "salu_insts_per_launch": sq.get("SQ_INSTS_SALU"), "l2_hit_rate": e["derived"]["l2_hit_rate"]}
    if sq.get("SQ_THREAD_CYCLES_VALU") and sq.get("SQ_ACTIVE_INST_VALU"):
        traffic[name]["valu_lane_utilisation"] = sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"] / 64.0
    if sq.get("SQ_WAIT_ANY") and sq.get("SQ_WAVE_CYCLES"):
        traffic[name]["wave_cycles_waiting"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
    e["derived"].update({k: traffic[name].get(k) for k in ("valu_lane_utilisation", "wave_cycles_waiting", "salu_insts_per_launch")})
    d = e["derived"]
    print(f"{key}: HBM {d['hbm_bytes_per_launch'] / 1e9:.1f} GB per launch = {d['hbm_GBps']:.0f} GB/s, L2 hit rate {d['l2_hit_rate']:.3f}, "
          f"VALU {d['valu_insts_per_launch']:.3g} per launch, loaded bytes {d['loads_bytes_per_launch']}")
json.dump(out, open(os.path.join(P, f"{tag}_pmc_workloads.json"), "w"), indent=1)
json.dump(traffic, open(tpath, "w"), indent=1)
