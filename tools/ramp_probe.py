#!/usr/bin/env python3
"""Developer measurement (under rocprofv3 --kernel-trace --stats): the fixed cost of one small
GEMM launch -- the same 1024-row x 1024-column output with K = 16, 64, 512, 1024, 2048."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pocketkaldi_amd as pk

rng = np.random.default_rng(0)
T, N = 1008, 1024
for K in (16, 64, 512, 1024, 2048):
    W = rng.standard_normal((N, K)).astype(np.float32)
    am = pk.AcousticModel([("linear", W, np.zeros(N, np.float32)), ("relu",),
                           ("linear", np.eye(16, N, dtype=np.float32), np.zeros(16, np.float32))], num_pdfs=16)
    x = rng.standard_normal((T, K)).astype(np.float32)
    for _ in range(30):
        am.propagate(x)
print("done")
