#!/usr/bin/env python3
"""Developer measurement: where the time of ONE 10 s utterance goes (BASELINE configs[1])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pocketkaldi_amd as pk
from pocketkaldi_amd import synth

layers, prior, L, R = synth.model("S")
am = pk.AcousticModel(layers, prior, L, R)
w = synth.utterance(0, 10.0)
bs = pk.BatchScorer(am, synth.global_cmvn_stats(), 1, len(w))
bs.set_waves([w])
for _ in range(20):
    bs.score(0.1, sync=True)
n = 300
t0 = time.perf_counter()
for _ in range(n):
    bs.score(0.1, sync=True)
sync_ms = (time.perf_counter() - t0) / n * 1e3
t0 = time.perf_counter()
for _ in range(n):
    bs.score(0.1, sync=False)
bs.synchronize()
async_ms = (time.perf_counter() - t0) / n * 1e3
bs.enable_timing(True)
bs.score(0.1, sync=True)
tm = bs.timing()
print("sync %.3f ms/utt, back-to-back async %.3f ms/utt; kernel events: %s (sum %.3f ms)"
      % (sync_ms, async_ms, {k: round(v[0], 4) for k, v in tm.items()}, sum(v[0] for v in tm.values())))
# pk_decodable_init path (host feats in, host log_prob out)
feats = bs.fetch_cmvn(0)
for _ in range(5):
    pk.Decodable(am, 0.1, feats).destroy()
t0 = time.perf_counter()
for _ in range(50):
    pk.Decodable(am, 0.1, feats).destroy()
print("pk_decodable_init (H2D feats, nnet, D2H 12 MB log_prob): %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))

if "--pairs" not in sys.argv:        # (under rocprofv3 the per-kernel averages should be the single-stream ones)
    sys.exit(0)
# Two independent utterances on two streams: do their kernels share the CUs and hide each other's
# launch ramps?  (the premise of splitting ONE utterance's frames into two pipelined halves)
bs2 = pk.BatchScorer(am, synth.global_cmvn_stats(), 1, len(w))
bs2.set_waves([synth.utterance(1, 10.0)])
bs.enable_timing(False)
for _ in range(10):
    bs.score(0.1, sync=False); bs2.score(0.1, sync=False)
bs.synchronize(); bs2.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    bs.score(0.1, sync=False); bs2.score(0.1, sync=False)
bs.synchronize(); bs2.synchronize()
print("two utterances in flight on two streams: %.3f ms per PAIR (one alone, back to back: %.3f ms each)"
      % ((time.perf_counter() - t0) / n * 1e3, async_ms))
t0 = time.perf_counter()
for _ in range(n):
    bs.score(0.1, sync=False); bs2.score(0.1, sync=False)
    bs.synchronize(); bs2.synchronize()
print("same, synchronised after every pair: %.3f ms per pair" % ((time.perf_counter() - t0) / n * 1e3))
