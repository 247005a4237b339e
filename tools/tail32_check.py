import os, sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import pocketkaldi_amd as pk
from pocketkaldi_amd import synth
from oracle import oracle as O
layers, prior, L, R = synth.model("S")
wave = synth.utterance(0, 10.0)
g = synth.global_cmvn_stats()
feats = O.cmvn(g, O.Fbank().compute(wave))
ref = O.Nnet(layers).am_compute(feats, prior, L, R, 0.1).astype(np.float64)
den = np.maximum(np.abs(ref), 1.0)
for flag in ("0", "1"):
    os.environ["PK_MI355_FUSED_TAIL32"] = flag
    am = pk.AcousticModel(layers, prior, L, R)
    got = pk.Decodable(am, 0.1, feats).log_prob()
    err = np.max(np.abs(got - ref) / den)
    bs = pk.BatchScorer(am, g, 1, len(wave))
    bs.set_waves([wave])
    for _ in range(20): bs.score(0.1, sync=True)
    t0 = time.perf_counter()
    for _ in range(300): bs.score(0.1, sync=True)
    ms = (time.perf_counter() - t0) / 300 * 1e3
    bs.enable_timing(True); bs.score(0.1, sync=True)
    print("PK_MI355_FUSED_TAIL32=%s: f32 err vs oracle %.2e; B=1 sync %.4f ms; tail event %.4f ms" % (flag, err, ms, bs.timing()["tail"][0]))
