#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the small, tracked summaries in profiles/.

  python profiles/summarize.py <tag> <stats dir> <pmc_fetch dir> <pmc_write dir> <points> [<kind> <frame> [<sq dir>...]]

Writes profiles/<tag>_kernel_stats.csv (per-kernel calls / avg / total from --kernel-trace
--stats), profiles/<tag>_pmc_traffic.csv (FETCH_SIZE / WRITE_SIZE per launch from the two
separate --pmc passes) and merges the per-launch byte counts into profiles/traffic.json under
the key "<kind>/<frame>/<points>" (what bench.py attaches as `traffic` for the SAME workload).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced streams (MI355X_MICROARCH.md, HBM section), so the read side is reported raw
AND doubled; the doubled figure is the one compared with algorithmic bytes.
"""
import collections
import csv
import glob
import json
import os
import sys

tag, stats_dir, fetch_dir, write_dir, points = sys.argv[1:6]
kind = sys.argv[6] if len(sys.argv) > 6 else "corridor"
frame = sys.argv[7] if len(sys.argv) > 7 else "offset"
sq_dirs = sys.argv[8:]
here = os.path.dirname(os.path.abspath(__file__))


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0].replace("pch::", "")


rows = list(csv.DictReader(open(glob.glob(os.path.join(stats_dir, "**/*kernel_stats.csv"), recursive=True)[0])))
ours = [r for r in rows if "pch::" in r["Name"]]
with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,avg_us,min_us,max_us,total_ms,percent_of_all_gpu_time\n")
    for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
        f.write(f"{short(r['Name'])},{r['Calls']},{float(r['AverageNs']) / 1e3:.2f},{float(r['MinNs']) / 1e3:.2f},"
                f"{float(r['MaxNs']) / 1e3:.2f},{float(r['TotalDurationNs']) / 1e6:.3f},{r['Percentage']}\n")


def pmc(d, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True)[0])):
        if r["Counter_Name"] == counter and "pch::" in r["Kernel_Name"]:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


fetch, write = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
steps_in_stats = 6                     # collect.sh: --steps 5 --warmup 1 in the kernel-trace run
steps_in_pmc = 3                       # --steps 2 --warmup 1 in the counter runs
names = {"ms_summary_k": "mean_summary", "ms_walk_k": "mean_walk", "ms_level2_k": "mean_level2",
         "db_core_k": "db_core", "db_core_k<false>": "db_core", "db_core_k<true>": "db_core_counting",
         "db_union_pairs_k": "db_union_pairs", "db_mark_k": "db_mark", "sel_bracket_k": "sel_bracket",
         "sg_hist_k": "seg_hist", "sg_scatter_k": "seg_scatter", "db_union_k": "db_union", "db_border_k": "db_border",
         "gf_compact_k<0>": "gf_compact", "gf_compact_k<1>": "gf_compact_fb", "db_chunksort_k": "db_chunksort", "sel_hist_k<0>": "sel_hist0",
         "sel_hist_k<1>": "sel_hist1", "sel_hist_k<2>": "sel_hist2", "rs_scatter_k": "radix_scatter",
         "rs_hist_k": "radix_hist", "db_gather_k": "db_gather", "db_keys_k": "db_keys",
         "sg_stats_k": "seg_stats", "db_label_k": "db_label", "db_cellbox_k": "db_cellbox",
         "db_union_face_k": "db_union0", "db_rowtab_k": "db_rowtab", "db_cellstats_k": "db_cellstats",
         "db_cells_k": "db_cells", "ms_sample_k": "mean_sample", "ms_prefix_k": "mean_prefix",
         "scan1_k<false>": "scan1", "scan1_k<true>": "scan1_popc", "db_prelabel_k": "db_prelabel",
         "sl_hist_k": "seg_hist", "sl_scatter_k": "seg_scatter", "sl_offsets_k": "seg_offsets",
         "vx_finish_k": "voxel_finish", "vx_scatter_k": "voxel_scatter", "vx_scatter_lds_k": "voxel_scatter",
         "vx_tilehist_k": "voxel_tilehist", "vx_minmax_k": "voxel_minmax", "vx_split_k": "voxel_split",
         "vx_binscan_k": "voxel_binscan"}
sys.path.insert(0, here)
from stamp import csrc_sha            # noqa: E402
traffic = {"points": int(points), "kind": kind, "frame": frame, "source": tag, "csrc_sha": csrc_sha(),
           "unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, KiB->B)", "kernels": {}}
with open(os.path.join(here, f"{tag}_pmc_traffic.csv"), "w") as f:
    f.write("kernel,launches,fetch_bytes_raw_per_launch,fetch_bytes_x2_per_launch,write_bytes_per_launch\n")
    for k in sorted(set(fetch) | set(write), key=lambda k: -sum(fetch.get(k, [0]))):
        fl, wl = fetch.get(k, []), write.get(k, [])
        fb = 1024 * sum(fl) / max(len(fl), 1)
        wb = 1024 * sum(wl) / max(len(wl), 1)
        f.write(f"{k},{max(len(fl), len(wl))},{fb:.0f},{2 * fb:.0f},{wb:.0f}\n")
        if k in names:
            traffic["kernels"][names[k]] = int(2 * fb + wb)
# per step: launches and bytes of EVERY kernel of the library (not only the named ones)
traffic["launches_per_step"] = round(sum(int(r["Calls"]) for r in ours) / steps_in_stats, 2)
traffic["kernel_ms_per_step"] = round(sum(float(r["TotalDurationNs"]) for r in ours) / 1e6 / steps_in_stats, 4)
tot = 0.0
for k in set(fetch) | set(write):
    tot += 1024 * (2 * sum(fetch.get(k, [])) + sum(write.get(k, [])))
traffic["step_bytes"] = int(tot / steps_in_pmc)
traffic["avg_us"] = {names.get(short(r["Name"]), short(r["Name"])): round(float(r["AverageNs"]) / 1e3, 2) for r in ours}
# SQ counters per launch (separate counters-only passes)
if sq_dirs:
    sq = collections.defaultdict(dict)
    for d in sq_dirs:
        files = glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(files[0])):
            if "pch::" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                sq[k][c] = sum(v) / max(len(v), 1)
    cols = sorted({c for v in sq.values() for c in v})
    with open(os.path.join(here, f"{tag}_sq_counters.csv"), "w") as f:
        f.write("kernel," + ",".join(cols) + ",avg_us,valu_issue_frac\n")
        dur = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in ours}
        for k in sorted(sq, key=lambda k: -sq[k].get("SQ_INSTS_VALU", 0)):
            us = dur.get(k, 0.0)
            # every VALU wave-instruction holds its SIMD's issue for >= 4 cycles (MI355X_MICROARCH.md, constants
            # table); 1024 SIMDs at <= 2.4 GHz.  A LOWER bound of the issue-slot occupancy.
            frac = sq[k].get("SQ_INSTS_VALU", 0) * 4 / (1024 * 2.4e9 * us * 1e-6) if us else 0.0
            f.write(k + "," + ",".join(f"{sq[k].get(c, 0):.0f}" for c in cols) + f",{us:.2f},{frac:.4f}\n")
            if k in names:
                traffic.setdefault("sq", {})[names[k]] = dict({c: int(sq[k].get(c, 0)) for c in cols},
                                                               valu_issue_frac=round(frac, 4))
tpath = os.path.join(here, "traffic.json")
try:
    allt = json.load(open(tpath))
    if "workloads" not in allt:
        allt = {"workloads": {}}
except Exception:
    allt = {"workloads": {}}
allt["workloads"][f"{kind}/{frame}/{int(points)}"] = traffic
json.dump(allt, open(tpath, "w"), indent=1)
print(open(os.path.join(here, f"{tag}_pmc_traffic.csv")).read())
