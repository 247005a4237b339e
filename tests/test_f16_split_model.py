"""CPU model of the f16x3 arithmetic (csrc/gemm_f16.hip: x = hi + lo, D = sum hi hi + hi lo + lo hi in fp32), numpy
float16 with subnormals, to pin down WHY the round-4 power-of-two prescale is needed and that it is exact:
  * unscaled, the error grows as the weights shrink (the lo halves fall into fp16 subnormals, then vanish);
  * prescaled so that max |W| is in [2^13, 2^14) -- what pk_mi355_am_finalize does (capi_model.hip) -- the error is
    the same at every scale, and the scaling itself changes no bit of the exact product.
No GPU, no product code: this is the arithmetic the GPU tests in test_gpu_f16_range.py then hold the kernels to.
"""
import numpy as np
import pytest


def split(x):
    x = np.clip(x.astype(np.float32), -65504.0, 65504.0)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def f16x3_matmul(X, W):
    xh, xl = split(X)
    wh, wl = split(W)
    return xh @ wh.T + xh @ wl.T + xl @ wh.T           # fp64 sums: isolates the operand error from the accumulation order


def finalize_exponent(W):
    """capi_model.hip, pk_mi355_am_finalize: 13 - ilogb(max |W|), clamped to +-60."""
    m = np.abs(W).max()
    return 0 if m == 0 else int(np.clip(13 - int(np.floor(np.log2(m))), -60, 60))


@pytest.mark.parametrize("log2_scale", [-16, -12, -8, -4, 0, 4])
def test_prescale_makes_the_error_scale_free(log2_scale):
    rng = np.random.default_rng(1)
    K = 1024
    W = (rng.standard_normal((64, K)) * np.sqrt(2.0 / K)).astype(np.float32) * np.float32(2.0 ** log2_scale)
    X = np.abs(rng.standard_normal((32, K))).astype(np.float32)
    exact = X.astype(np.float64) @ W.astype(np.float64).T
    ref = np.abs(exact).max()
    plain = np.abs(f16x3_matmul(X, W) - exact).max() / ref
    e = finalize_exponent(W)
    Ws = W * np.float32(2.0 ** e)
    assert np.array_equal(Ws.astype(np.float64) * 2.0 ** -e, W.astype(np.float64))      # exact both ways
    assert 2.0 ** 13 <= np.abs(Ws).max() < 2.0 ** 14
    scaled = np.abs(f16x3_matmul(X, Ws) * 2.0 ** -e - exact).max() / ref
    assert scaled < 1e-6
    if log2_scale <= -8:
        assert plain > 20 * scaled          # what round 3 shipped: 1e-4 .. 1e-3 here


def test_the_error_floor_is_the_dropped_lo_lo_term():
    rng = np.random.default_rng(2)
    K = 2048
    W = (rng.standard_normal((32, K)) * 2.0 ** 13 / 4).astype(np.float32)
    X = (rng.standard_normal((16, K)) * 256).astype(np.float32)
    exact = X.astype(np.float64) @ W.astype(np.float64).T
    err = np.abs(f16x3_matmul(X, W) - exact).max() / np.abs(exact).max()
    assert err < 2.0 ** -20
