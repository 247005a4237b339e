#!/usr/bin/env python3
"""Developer check: the one-launch layer stack (PK_MI355_NET=1) against the multi-launch path, same process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import pocketkaldi_amd as pk
from pocketkaldi_amd import synth

def run(model, T, D=40, seed=0):
    layers, prior, L, R = synth.model(model)
    rng = np.random.default_rng(seed)
    feats = rng.standard_normal((T, D)).astype(np.float32)
    out = {}
    for net in ("0", "1"):
        os.environ["PK_MI355_NET"] = net
        am = pk.AcousticModel(layers, prior, L, R).set_softmax("reference")
        d = pk.Decodable(am, 0.1, feats)
        out[net] = d.log_prob().copy()
        d.destroy()
    a, b = out["0"].view(np.uint32), out["1"].view(np.uint32)
    bad = a != b
    print("model %s T %d: %d / %d elements differ; rows with differences: %s; cols: %s" % (
        model, T, bad.sum(), bad.size, np.unique(np.nonzero(bad)[0] // 64)[:20], np.unique(np.nonzero(bad)[1] // 32)[:20]), flush=True)
    return bad.sum()

for model, T in (("tiny", 47), ("tiny", 300), ("tiny", 300), ("tiny", 1000), ("S", 300), ("S", 998), ("S", 998)):
    run(model, T)

# random stacks: every pair must be bit-identical (NaN patterns included)
rng = np.random.default_rng(12345)
fails = 0
for case in range(int(os.environ.get("NET_FUZZ", "150"))):
    D = int(rng.integers(1, 49)); L = int(rng.integers(0, 6)); R = int(rng.integers(0, 6))
    depth = int(rng.integers(1, 6))
    dims = [D * (L + R + 1)] + [int(rng.integers(1, 1100)) for _ in range(depth)]
    layers = []
    for i in range(depth):
        layers.append(("linear", (rng.standard_normal((dims[i + 1], dims[i])) * np.sqrt(2.0 / dims[i])).astype(np.float32),
                       (rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32)))
        if i < depth - 1 or rng.random() < 0.3:
            layers.append(("relu",))
    if rng.random() < 0.7:
        layers.append(("softmax",))
    N = dims[-1]
    prior = rng.uniform(0.5, 1.5, N); prior = (prior / prior.sum()).astype(np.float32)
    T = int(rng.choice([1, 5, 60, 64, 65, 300, 500, 998, 1500, 3000]))
    feats = rng.standard_normal((T, D)).astype(np.float32)
    outs = []
    for net in ("0", "1"):
        os.environ["PK_MI355_NET"] = net
        am = pk.AcousticModel(layers, prior, L, R).set_softmax("reference")
        d = pk.Decodable(am, 0.1, feats)
        outs.append(d.log_prob().copy()); d.destroy()
    same = np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    if not same:
        fails += 1
        print("MISMATCH case %d dims %s L %d R %d T %d layers %s: %d elements" % (
            case, dims, L, R, T, [l[0] for l in layers], (outs[0].view(np.uint32) != outs[1].view(np.uint32)).sum()), flush=True)
print("fuzz: %d mismatching cases" % fails)
