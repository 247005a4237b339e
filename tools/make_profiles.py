#!/usr/bin/env python3
"""Turn the raw output of tools/profile_gpu.sh (gpurun_out/prof/) into the committed summaries
under profiles/:  usage: tools/make_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
SRC = "gpurun_out/prof"
DST = "profiles"


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


def short(name):
    m = re.search(r"(\w+Kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def counters(d):
    """{(kernel, grid): {counter: [values per dispatch]}} of one PMC pass"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(one(d + "/**/*counter_collection.csv"))):
        per[(r["Dispatch_Id"], short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]] += float(r["Counter_Value"])
    for (_, k, g), c in per.items():
        for name, v in c.items():
            agg[(k, g)][name].append(v)
    return agg


for mode in ("f32", "f16x3"):
    shutil.copy(one(mode + "/**/*kernel_stats.csv"), os.path.join(DST, "%s_%s_kernel_stats.csv" % (tag, mode)))
    line = open(os.path.join(SRC, mode + "_bench.json")).read().strip().splitlines()[-1]
    json.dump(json.loads(line), open(os.path.join(DST, "%s_%s_bench_under_rocprof.json" % (tag, mode)), "w"), indent=1)
line = open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1]
json.dump(json.loads(line), open(os.path.join(DST, tag + "_bench.json"), "w"), indent=1)

fetch, write = counters("pmc_fetch"), counters("pmc_write")
kernels, gemm_bytes, gemm_n = [], 0.0, 0
for key in sorted(fetch):
    f = fetch[key]["FETCH_SIZE"]
    w = write.get(key, {}).get("WRITE_SIZE", [0.0])
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    hbm = 2.0 * fk * 1024 + wk * 1024
    kernels.append({"kernel": key[0], "grid_threads": key[1], "dispatches_sampled": len(f),
                    "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch_corrected": hbm})
    if key[0].startswith("GemmKernel"):
        gemm_bytes += hbm * len(f)
        gemm_n += len(f)
json.dump({
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 1 --warmup 1 "
            "--no-cpu-baseline --no-other-precision --no-other-configs` (model S, 256 utt x 10 s, f32; tools/profile_gpu.sh). "
            "Counters are in KB. gfx950 correction per MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-B requests at "
            "64 B for wide coalesced reads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.",
    "kernels": kernels, "gemm_avg_hbm_bytes_per_launch": gemm_bytes / max(gemm_n, 1), "gemm_launches_sampled": gemm_n},
    open(os.path.join(DST, tag + "_pmc_traffic.json"), "w"), indent=1)

out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE ... (own pass, tools/profile_gpu.sh). "
               "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES * 4 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs.",
       "f32": [], "f16x3": []}
for mode, d in (("f32", "pmc_mfma"), ("f16x3", "pmc_mfma_f16")):
    for key, c in sorted(counters(d).items()):
        if "Gemm" not in key[0]:
            continue
        avg = {k: sum(v) / len(v) for k, v in c.items()}
        row = {"kernel": key[0], "grid_threads": key[1], "dispatches_sampled": len(next(iter(c.values())))}
        row.update(avg)
        if avg.get("SQ_BUSY_CU_CYCLES"):
            row["mfma_busy"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["SQ_BUSY_CU_CYCLES"] * 4.0)
        out[mode].append(row)
json.dump(out, open(os.path.join(DST, tag + "_pmc_mfma.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(DST)))
