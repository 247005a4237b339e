import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import pocketkaldi_amd as pk
from pocketkaldi_amd import synth
B = 256
layers, prior, L, R = synth.model("S")
am = pk.AcousticModel(layers, prior, L, R)
waves = [synth.utterance(u, 10.0).astype(np.int16) for u in range(B)]
tot = sum(len(w) for w in waves)
A = pk.BatchScorer(am, synth.global_cmvn_stats(), B, tot)
Bb = pk.BatchScorer(am, synth.global_cmvn_stats(), B, tot)
for s in (A, Bb):
    s.set_waves_i16(waves); s.score(0.1, sync=True); s.fetch_all()
def t(f):
    t0 = time.perf_counter(); f(); return (time.perf_counter() - t0) * 1e3
print("lone copy %.1f ms" % t(lambda: A.fetch_all()))
print("lone score %.1f ms" % t(lambda: Bb.score(0.1, sync=True)))
def both():
    for _ in range(4): Bb.score(0.1, sync=False)
    t0 = time.perf_counter(); A.fetch_all(sync=True); t1 = time.perf_counter()
    Bb.synchronize(); t2 = time.perf_counter()
    print("  copy under 4 scores: copy done after %.1f ms, scores done after %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
both(); both()
def both2():
    t0 = time.perf_counter()
    A.fetch_all(sync=False)
    for _ in range(2): Bb.score(0.1, sync=False)
    t1 = time.perf_counter()
    A.synchronize(); t2 = time.perf_counter()
    Bb.synchronize(); t3 = time.perf_counter()
    print("  copy first then 2 scores: enqueue %.1f ms, copy done %.1f ms, scores done %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t0) * 1e3))
both2(); both2()

cat = np.concatenate(waves); ns = [len(w) for w in waves]
pin = pk.pinned_i16(len(cat)); pin[:] = cat
for name, src in (("pageable", cat), ("pinned", pin)):
    Bb.synchronize(); A.synchronize()
    print("%s upload alone: %.1f ms" % (name, t(lambda: Bb.set_waves_i16_raw(src, ns))))
    A.fetch_all(sync=False)
    print("%s upload while another batch's D2H runs: %.1f ms" % (name, t(lambda: Bb.set_waves_i16_raw(src, ns))))
    t0 = time.perf_counter(); Bb.score(0.1, sync=True); t1 = time.perf_counter(); A.synchronize(); t2 = time.perf_counter()
    print("   then score done after %.1f ms, copy done after %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))

print("-- pipeline trace (ms since start): sync_done, upload_done, score_enq, fetch_enq")
pair = [A, Bb]
pending = [False, False]
A.synchronize(); Bb.synchronize()
T0 = time.perf_counter()
now = lambda: (time.perf_counter() - T0) * 1e3
for k in range(8):
    s = pair[k % 2]
    a = now()
    if pending[k % 2]:
        s.synchronize()
    b = now()
    s.set_waves_i16_raw(pin, ns)
    c = now()
    s.score(0.1, sync=False)
    d = now()
    s.fetch_all(sync=False)
    e = now()
    pending[k % 2] = True
    print("k=%d  wait %.1f->%.1f  upload %.1f  score_enq %.1f  fetch_enq %.1f" % (k, a, b, c, d, e), flush=True)
A.synchronize(); Bb.synchronize()
print("all done %.1f" % now())
