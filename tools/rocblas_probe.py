#!/usr/bin/env python3
"""Developer measurement: what the vendor library's fp32 GEMM does on the hidden-layer shape
(65 536 x 1024 x 1024), for context next to GemmKernel.  Not part of the product or of bench.py."""
import time
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)
for (M, K, N) in [(65536, 1024, 1024), (65536, 1024, 3072), (65536, 448, 1024)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float32)
    b = torch.randn(K, N, device=dev, dtype=torch.float32)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("torch.mm fp32 %d x %d x %d: %.3f ms, %.1f TFLOP/s" % (M, K, N, ms, 2.0 * M * K * N / ms / 1e9), flush=True)
