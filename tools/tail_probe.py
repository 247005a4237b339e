#!/usr/bin/env python3
"""Developer probe: TailKernel time against rows per launch (the layer-stack chunk) on a model that is
nothing but one affine layer + softmax, 256 x 10 s.  usage: tools/tail_probe.py [chunk ...]"""
import os
import sys

import numpy as np
import torch  # noqa: F401  (first: one HIP runtime in the process)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pocketkaldi_amd as pk
from pocketkaldi_amd import synth

chunks = [int(a) for a in sys.argv[1:]] or [131072, 126976, 129024, 65536, 64512, 43008, 32768]
rng = np.random.default_rng(0)
N = int(os.environ.get("PROBE_PDFS", "3000"))
layers = [("linear", (rng.standard_normal((N, 440)) * 0.05).astype(np.float32), np.zeros(N, np.float32)), ("softmax",)]
prior = np.full(N, 1.0 / N, np.float32)
g = synth.global_cmvn_stats()
waves = [synth.utterance(u, 10.0).astype(np.int16) for u in range(8)] * 32
for c in chunks:
    os.environ["PK_MI355_CHUNK"] = str(c)
    am = pk.AcousticModel(layers, prior, 5, 5)
    bs = pk.BatchScorer(am, g, len(waves), sum(len(w) for w in waves))
    bs.set_waves_i16(waves)
    for _ in range(2):
        bs.score(0.1)
    bs.enable_timing(True)
    ts = []
    for _ in range(4):
        bs.score(0.1)
        ts.append(bs.timing()["tail"])
    ms = min(t[0] for t in ts)
    rows = 256 * 1008
    print("chunk %7d  launches %d  tail %.3f ms/step  %.2f ns/row  %.2f TB/s" % (
        c, ts[0][1], ms, ms * 1e6 / rows, rows * N * 8 / ms / 1e9), flush=True)
    del bs, am
