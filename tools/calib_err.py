import sys, numpy as np
sys.path.insert(0, '/root/repo')
import pocketkaldi_amd as pk
from pocketkaldi_amd import synth
from oracle import oracle as O
for name in ("S", "W"):
    layers, prior, L, R = synth.model(name)
    wave = synth.utterance(0, 2.0)
    g = synth.global_cmvn_stats()
    feats = O.cmvn(g, O.Fbank().compute(wave))
    ref = O.Nnet(layers).am_compute(feats, prior, L, R, 0.1).astype(np.float64)
    den = np.maximum(np.abs(ref), 1.0)
    am = pk.AcousticModel(layers, prior, L, R, precision="f16x3")
    e0 = np.max(np.abs(pk.Decodable(am, 0.1, feats).log_prob() - ref) / den)
    am.calibrate(feats)
    e1 = np.max(np.abs(pk.Decodable(am, 0.1, feats).log_prob() - ref) / den)
    print(name, "f16x3 err vs oracle: uncalibrated %.2e, calibrated %.2e; exponents" % (e0, e1), am.exponents())
    am32 = pk.AcousticModel(layers, prior, L, R)
    print(name, "f32 (stable tail) err vs oracle %.2e" % np.max(np.abs(pk.Decodable(am32, 0.1, feats).log_prob() - ref) / den))
