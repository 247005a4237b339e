#!/usr/bin/env python3
"""Developer measurement: is GemmF16K32Kernel limited by its instruction stream or by power?
The same launches (model W, f16x3, 256 x 10 s) with three weight sets -- the same instructions, different bit
toggling in the matrix pipes: He-normal (the benchmark's), all zeros, all ones (one value: operands toggle, products do not)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pocketkaldi_amd as pk
from pocketkaldi_amd import synth

model = sys.argv[1] if len(sys.argv) > 1 else "W"
layers, prior, L, R = synth.model(model)
waves = [synth.utterance(u, 10.0) for u in range(256)]
ns = [len(w) for w in waves]
pcm = torch.from_numpy(np.concatenate(waves)).to("cuda:0")
g = synth.global_cmvn_stats()
for name, f in (("he-normal", lambda w: w), ("zeros", lambda w: np.zeros_like(w)), ("ones/64", lambda w: np.full_like(w, 1.0 / 64)),
                ("he-normal again", lambda w: w)):
    ls = [(l[0], f(l[1]), f(l[2]) if name != "he-normal" and name != "he-normal again" else l[2]) if l[0] == "linear" else l for l in layers]
    am = pk.AcousticModel(ls, prior, L, R, precision="f16x3")
    bs = pk.BatchScorer(am, g, 256, int(sum(ns)))
    bs.set_waves_device(pcm.data_ptr(), ns, keep_alive=pcm)
    try:
        bs.calibrate()
    except pk.PkError as e:
        print(name, "calibrate:", e)
    for _ in range(2):
        bs.score(0.1, sync=False)
    bs.synchronize() if name.startswith("he") else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(6):
        if k == 5:
            bs.enable_timing(True)
        bs.score(0.1, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    tm = bs.timing()
    gemm_ms = tm["gemm"][0]
    tf = am.flops_per_frame() * bs.total_frames() / (gemm_ms * 1e-3) / 1e12
    print("%-16s ms/step %.2f  gemm %.2f ms = %.1f TFLOP/s algorithmic  (tail %.2f)" % (name, dt * 1e3, gemm_ms, tf, tm["tail"][0]), flush=True)
    bs.close(); am.close()
