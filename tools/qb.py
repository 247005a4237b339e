#!/usr/bin/env python3
"""Developer helper: run bench.py under several env settings, print the key numbers.
usage: tools/qb.py [bench args] -- NAME=V1,V2,...   (one swept env var)"""
import json
import os
import subprocess
import sys

args = sys.argv[1:]
sweep = None
if "--" in args:
    i = args.index("--")
    sweep = args[i + 1]
    args = args[:i]
name, vals = (sweep.split("=")[0], sweep.split("=")[1].split(",")) if sweep else ("_", [""])
for v in vals:
    env = dict(os.environ)
    if sweep:
        env[name] = v
    p = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-other-precision", "--no-other-configs"] + args, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        st = d["stage_ms_per_step"]
        print("%s=%s  frames/s %.3e  ms/step %.2f  gemm %.1f TF (%.3f)  stages: %s" % (
            name, v, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"],
            " ".join("%s=%.2f" % kv for kv in st.items())), flush=True)
    except Exception as e:
        print(name, v, "FAILED", e, p.stderr[-2000:], flush=True)
