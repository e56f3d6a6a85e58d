#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes written by scripts/pmc_gpu.sh into gpurun_out/pmc_<tag>.json (per kernel: mean
counter values per launch + derived HBM traffic, MFMA pipe busy fraction, L2 hit rate).  FETCH_SIZE is doubled as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on gfx950; FETCH/WRITE are KB."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in agg.items():
    if not any(s in k for s in ("corr_", "pgo", "kp_", "match_cov", "lookup", "upsample", "patch_embed_kernel", "patch_embed_strip_kernel", "patch_embed_pipelined_kernel", "backend_front")):
        continue
    name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip() or k[:60]
    m = {c: sum(x) / len(x) for c, x in sorted(v.items())}
    d = {"launch_config": f"tools/kernel_bench.py {cfg} --iters 5", "launches_sampled": max(len(x) for x in v.values())}
    if "FETCH_SIZE" in m:
        d["hbm_read_bytes_uncorrected"] = m["FETCH_SIZE"] * 1024
        d["hbm_read_bytes_gfx950_corrected_x2"] = 2 * m["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in m:
        d["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        d["traffic_bytes_per_launch"] = 2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # busy cycles summed over 1024 SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs -> kernel cycles = GUI / 8
        d["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8)
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        d["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    m["_derived"] = d
    out[name] = m
    print(name)
    for c, val in m.items():
        if c != "_derived":
            print("    %-34s mean=%.4e" % (c, val))
    print("    derived:", {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d.items()})
json.dump(out, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
