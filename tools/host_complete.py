#!/usr/bin/env python3
"""Developer measurement: the PCIe-inclusive ("host-complete") rate of the batch path --
host PCM in, every utterance's log-likelihoods copied into a host pk_decodable_t.
Not the benchmark metric (bench.py measures device-complete); quoted in DESIGN.md."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pocketkaldi_amd as pk
from pocketkaldi_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layers, prior, L, R = synth.model("S")
am = pk.AcousticModel(layers, prior, L, R)
waves = [synth.utterance(u, 10.0) for u in range(B)]
bs = pk.BatchScorer(am, synth.global_cmvn_stats(), B, sum(len(w) for w in waves))
for rep in range(3):
    t0 = time.perf_counter()
    bs.set_waves(waves)                      # H2D, pageable host memory
    t1 = time.perf_counter()
    bs.score(0.1, sync=True)
    t2 = time.perf_counter()
    ds = [bs.fetch(u) for u in range(B)]     # D2H into malloc'd log_prob, one per utterance
    t3 = time.perf_counter()
    frames = bs.total_frames()
    print("rep %d: H2D %.1f ms, score %.1f ms, D2H %.1f ms (%.1f GB/s) -> host-complete %.3g frames/s, device-complete %.3g"
          % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, frames * 3000 * 4 / (t3 - t2) / 1e9,
             frames / (t3 - t0), frames / (t2 - t1)), flush=True)
    for d in ds:
        d.destroy()

# Page-locked arena + one transfer per batch (pk_mi355_batch_fetch_all), two batches in flight
# so that one batch's D2H runs while the other is being scored.
print("-- fetch_all, two batches in flight", flush=True)
pair = [bs, pk.BatchScorer(am, synth.global_cmvn_stats(), B, sum(len(w) for w in waves))]
w16 = [w.astype(np.int16) for w in waves]
for s in pair:                                   # first use allocates the arena: untimed
    s.set_waves_i16(w16); s.score(0.1, sync=True); s.fetch_all()
t0 = time.perf_counter()
pair[0].fetch_all()
dt = time.perf_counter() - t0
print("one fetch_all alone (idle GPU): %.1f ms = %.1f GB/s" % (dt * 1e3, pair[0].total_frames() * 3000 * 4 / dt / 1e9), flush=True)
for rep in range(3):
    n_batches = 16
    t0 = time.perf_counter()
    pending = [None, None]
    done = 0
    for k in range(n_batches + 2):
        s = pair[k % 2]
        if pending[k % 2] is not None:
            s.synchronize()                      # this batch's log-likelihoods are on the host now
            done += s.total_frames()
            pending[k % 2] = None
        if k < n_batches:
            s.set_waves_i16(w16)                 # int16 PCM, 320 B/frame over the link
            s.score(0.1, sync=False)
            pending[k % 2] = s.fetch_all(sync=False)
    dt = time.perf_counter() - t0
    print("rep %d: %d batches, %.1f ms each -> host-complete %.3g frames/s (%.1f GB/s D2H average)"
          % (rep, n_batches, dt / n_batches * 1e3, done / dt, done * 3000 * 4 / dt / 1e9), flush=True)
