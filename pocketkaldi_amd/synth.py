"""Seeded synthetic inputs shared by the tests and bench.py.

There is no network, so audio, CMVN statistics, weights and priors are synthetic:
16 kHz mono PCM as pk_16kpcm_read hands it over (integer-valued floats,
pcm_reader.cc:189-211), He-normal weights, normalised uniform priors.
"""
import numpy as np

SAMPLE_RATE = 16000

# name -> (hidden width, hidden layers, pdfs): BASELINE.json configs 2-4 (S) and 5 (W)
MODELS = {"S": (1024, 4, 3000), "W": (2048, 6, 8000), "tiny": (64, 2, 50)}


def utterance(utt_id, seconds=10.0, dtype=np.float32):
    rng = np.random.default_rng(0xACE0 + int(utt_id))
    n = int(round(seconds * SAMPLE_RATE))
    t = np.arange(n, dtype=np.float64)
    f1 = rng.uniform(80, 400)
    f2 = rng.uniform(500, 3500)
    phi = rng.uniform(0, 2 * np.pi)
    s = (3000 * np.sin(2 * np.pi * f1 * t / SAMPLE_RATE)
         + 2000 * np.sin(2 * np.pi * f2 * t / SAMPLE_RATE + phi)
         + 1500 * rng.standard_normal(n))
    return np.round(np.clip(s, -32767, 32767)).astype(dtype)


def ragged_seconds(utt_id, lo=2.0, hi=20.0):
    """Length of utterance `utt_id` in bench.py's ragged workload: U[lo, hi] seconds, seeded by the id
    alone (every rank computes the whole list), rounded to a whole sample."""
    s = np.random.default_rng(0xBA5E0 + int(utt_id)).uniform(lo, hi)
    return round(s * SAMPLE_RATE) / SAMPLE_RATE


def global_cmvn_stats():
    gn = 1.0e6
    g = np.empty(41, dtype=np.float32)
    g[:40] = gn * (12.0 + 0.1 * np.arange(40))
    g[40] = gn
    return g


def model(name="S", feat_dim=40, left=5, right=5, seed=0x5EED):
    """-> (layers, prior, left, right) in the form AcousticModel / oracle.Nnet take."""
    hidden, nh, pdfs = MODELS[name]
    rng = np.random.default_rng(seed)
    dims = [feat_dim * (left + right + 1)] + [hidden] * nh + [pdfs]
    layers = []
    for i in range(len(dims) - 1):
        fan_in, fan_out = dims[i], dims[i + 1]
        W = (rng.standard_normal((fan_out, fan_in)) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        b = (rng.standard_normal(fan_out) * 0.1).astype(np.float32)
        layers.append(("linear", W, b))
        layers.append(("relu",) if i < len(dims) - 2 else ("softmax",))
    prior = rng.uniform(0.5, 1.5, pdfs)
    prior = (prior / prior.sum()).astype(np.float32)
    return layers, prior, left, right
