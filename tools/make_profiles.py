#!/usr/bin/env python3
"""Turn the raw output of tools/profile_gpu.sh (gpurun_out/prof/) into the committed summaries
under profiles/:   usage: tools/make_profiles.py r02

Per section found (S = headline workload, W = BASELINE configs[4] wide model, B1 = configs[1]):
  rNN_<sec>_<mode>_kernel_stats.csv        rocprofv3 --kernel-trace --stats summary, verbatim
  rNN_<sec>_<mode>_bench_under_rocprof.json the bench line of that same run
  rNN_<sec>_pmc_traffic.json               FETCH_SIZE / WRITE_SIZE passes -> HBM bytes per launch
  rNN_<sec>_pmc_mfma.json                  MFMA-busy passes (+ effective clock from GRBM_GUI_ACTIVE)
  rNN_B1_kernel_stats.csv, rNN_B1_probe.txt
  rNN_bench.json                           the plain benchmark line
"""
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
ALLOW_MIXED = "--allow-mixed" in sys.argv


def section_head(sec):
    """{library_build_hash, gemm_source_hash} tools/profile_gpu.sh recorded when it ran section `sec`."""
    path = os.path.join(SRC, sec + ".head")
    if not os.path.exists(path):
        return {"library_build_hash": None, "gemm_source_hash": None}
    return dict(kv.split("=", 1) for kv in open(path).read().split())


def heads_check():
    """All sections present must have been measured on ONE library (profile_gpu.sh re-runs
    single sections, and a summary mixing HEADs under one tag is silently wrong)."""
    seen = {s: section_head(s)["library_build_hash"] for s in ("S", "W", "B1", "bench")
            if os.path.exists(os.path.join(SRC, s + ".head"))}
    if len(set(seen.values())) > 1:
        msg = "sections measured on different library builds: %s" % seen
        if not ALLOW_MIXED:
            raise SystemExit(msg + "  (re-run the stale sections, or pass --allow-mixed: every file then carries its own hash)")
        print("WARNING:", msg)
    return seen


heads_check()
try:
    import subprocess
    GIT_HEAD = subprocess.check_output(["git", "rev-parse", "HEAD"], text=True).strip()
except Exception:
    GIT_HEAD = None


def one(pattern, required=True):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not hits:
        if required:
            raise SystemExit("missing " + pattern)
        return None
    return hits[0]


def short(name):
    m = re.search(r"(\w+Kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def counters(d):
    """{(kernel, grid): {counter: [values per dispatch]}, ...} and per-dispatch durations of one PMC pass"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    for r in csv.DictReader(open(one(d + "/**/*counter_collection.csv"))):
        key = (r["Dispatch_Id"], short(r["Kernel_Name"]), int(r["Grid_Size"]))
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    durs = collections.defaultdict(list)
    for (did, k, g), c in per.items():
        for name, v in c.items():
            agg[(k, g)][name].append(v)
        durs[(k, g)].append(dur[(did, k, g)])
    return agg, durs


def bench_line(path):
    line = open(path).read().strip().splitlines()[-1]
    return json.loads(line)


wrote = []
for sec in ("S", "W"):
    for mode in ("f32", "f16x3"):
        stats = one("%s_%s/**/*kernel_stats.csv" % (sec, mode), required=False)
        if not stats:
            continue
        shutil.copy(stats, os.path.join(DST, "%s_%s_%s_kernel_stats.csv" % (tag, sec, mode)))
        line = bench_line(os.path.join(SRC, "%s_%s_bench.json" % (sec, mode)))
        line["measured_on"] = dict(section_head(sec), summarised_at_git_head=GIT_HEAD)
        json.dump(line, open(os.path.join(DST, "%s_%s_%s_bench_under_rocprof.json" % (tag, sec, mode)), "w"), indent=1)
        wrote += ["%s_%s kernel stats" % (sec, mode)]
    # HBM traffic per launch
    head = section_head(sec)
    head["summarised_at_git_head"] = GIT_HEAD
    traffic = {"measured_on": head, "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 1 --warmup 1 "
                       "--no-cpu-baseline --no-other-precision --no-other-configs --no-host-endpoints%s [--precision f16x3]` "
                       "(256 utt x 10 s; tools/profile_gpu.sh). Counters are in KB. gfx950 correction per MI355X_MICROARCH.md (HBM "
                       "section): FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads, so read bytes = 2 x FETCH_SIZE x "
                       "1024; WRITE_SIZE is exact." % (" --model W" if sec == "W" else "")}
    mfma = {"measured_on": head, "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY "
                    "SQ_LDS_BANK_CONFLICT (own pass, tools/profile_gpu.sh). mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES "
                    "* 4 SIMDs); effective_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md, DVFS "
                    "give-back); mfma_rate_frac_of_2p4ghz_peak = mfma_busy * clock / 2.4."}
    have = False
    for mode in ("f32", "f16x3"):
        if not one("%s_pmc_fetch_%s/**/*counter_collection.csv" % (sec, mode), required=False):
            continue
        have = True
        fetch, _ = counters("%s_pmc_fetch_%s" % (sec, mode))
        write, _ = counters("%s_pmc_write_%s" % (sec, mode))
        kernels, gemm_bytes, gemm_n = [], 0.0, 0
        for key in sorted(fetch):
            f = fetch[key]["FETCH_SIZE"]
            w = write.get(key, {}).get("WRITE_SIZE", [0.0])
            fk, wk = sum(f) / len(f), sum(w) / len(w)
            hbm = 2.0 * fk * 1024 + wk * 1024
            kernels.append({"kernel": key[0], "grid_threads": key[1], "dispatches_sampled": len(f),
                            "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch_corrected": hbm})
            if key[0].startswith("Gemm"):
                gemm_bytes += hbm * len(f)
                gemm_n += len(f)
        traffic[mode] = {"kernels": kernels, "gemm_avg_hbm_bytes_per_launch": gemm_bytes / max(gemm_n, 1),
                         "gemm_launches_sampled": gemm_n}
        agg, durs = counters("%s_pmc_mfma_%s" % (sec, mode))
        rows = []
        for key, c in sorted(agg.items()):
            if "Gemm" not in key[0]:
                continue
            avg = {k: sum(v) / len(v) for k, v in c.items()}
            row = {"kernel": key[0], "grid_threads": key[1], "dispatches_sampled": len(next(iter(c.values()))),
                   "avg_duration_us": sum(durs[key]) / len(durs[key]) * 1e6}
            row.update(avg)
            if avg.get("SQ_BUSY_CU_CYCLES"):
                row["mfma_busy"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["SQ_BUSY_CU_CYCLES"] * 4.0)
            if avg.get("GRBM_GUI_ACTIVE"):
                row["effective_clock_ghz"] = avg["GRBM_GUI_ACTIVE"] / 8.0 / (row["avg_duration_us"] * 1e-6) / 1e9
                if "mfma_busy" in row:
                    row["mfma_rate_frac_of_2p4ghz_peak"] = row["mfma_busy"] * row["effective_clock_ghz"] / 2.4
            rows.append(row)
        mfma[mode] = rows
    if have:
        if sec == "S" and "f32" in traffic:       # bench.py reads this key
            traffic["gemm_avg_hbm_bytes_per_launch"] = traffic["f32"]["gemm_avg_hbm_bytes_per_launch"]
        json.dump(traffic, open(os.path.join(DST, "%s_%s_pmc_traffic.json" % (tag, sec)), "w"), indent=1)
        json.dump(mfma, open(os.path.join(DST, "%s_%s_pmc_mfma.json" % (tag, sec)), "w"), indent=1)
        wrote += ["%s pmc" % sec]

b1 = one("B1/**/*kernel_stats.csv", required=False)
if b1:
    shutil.copy(b1, os.path.join(DST, tag + "_B1_kernel_stats.csv"))
    keep = [l for l in open(os.path.join(SRC, "B1_probe.log")) if not re.match(r"^[EWI]\d{8} ", l) and not l.startswith("/opt")]
    open(os.path.join(DST, tag + "_B1_probe.txt"), "w").write("".join(keep))
    wrote += ["B1"]
if os.path.exists(os.path.join(SRC, "bench.json")):
    line = bench_line(os.path.join(SRC, "bench.json"))
    line["measured_on"] = dict(section_head("bench"), summarised_at_git_head=GIT_HEAD)
    json.dump(line, open(os.path.join(DST, tag + "_bench.json"), "w"), indent=1)
    wrote += ["bench"]
# bench.py's roofline.traffic reads profiles/rNN_pmc_traffic.json (newest round first)
if os.path.exists(os.path.join(DST, tag + "_S_pmc_traffic.json")):
    shutil.copy(os.path.join(DST, tag + "_S_pmc_traffic.json"), os.path.join(DST, tag + "_pmc_traffic.json"))
print("wrote", wrote, sorted(f for f in os.listdir(DST) if f.startswith(tag)))
