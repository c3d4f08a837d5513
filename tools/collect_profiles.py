#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into profiles/<tag>/."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", tag)
dst = os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in ("bench.json", "bench_under_rocprof.json", "bench_rccl_1rank.json", "ablation.txt", "ubench.txt", "ubench3.txt", "quality_bench.json",
          "pipeline_and_materialize.json", "path_sweep.txt", "path_pmc.txt", "min_grid.txt", "compat_bench.txt", "pytest_gpu.log", "gpu_fuzz.log", "wbw.txt", "min_generic.txt", "min_ab.txt", "min_generic_contigs.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
ks = os.path.join(src, "trace", "p_kernel_stats.csv")
if os.path.exists(ks):
    shutil.copy(ks, os.path.join(dst, "rocprofv3_kernel_stats.csv"))
pmc = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3"):
    p = os.path.join(src, d, "p_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "scan_kernel" in r["Kernel_Name"] or "scan2_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k] = {"dispatches": len(v), "mean_per_dispatch": sum(v) / len(v)}
if "FETCH_SIZE" in pmc:
    # rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KB; on gfx950 FETCH_SIZE is exactly half of the bytes of a wide
    # coalesced streaming read (MI355X_MICROARCH.md, HBM section) -> double it.
    pmc["hbm_read_bytes_per_launch_corrected"] = pmc["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2
if "WRITE_SIZE" in pmc:
    pmc["hbm_write_bytes_per_launch"] = pmc["WRITE_SIZE"]["mean_per_dispatch"] * 1024
json.dump(pmc, open(os.path.join(dst, "pmc_scan_kernel.json"), "w"), indent=1)
print(json.dumps(pmc, indent=1)[:1500])
