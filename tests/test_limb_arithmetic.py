"""CPU checks of the int8-limb (block-fixed-point) GEMM emulation in scripts/limb_precision_study.py - the arithmetic a
future int8-limb dW / reverse sweep would implement: limb ranges fit a signed byte, the
limbs reconstruct the quantised integer exactly, the three-term product is the four-term one minus a0*b0, the sums fit
an i32 accumulator at K = 256, and the bf16 rounding helper is torch's."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import limb_precision_study as lp      # noqa: E402


def test_limbs_fit_a_byte_and_reconstruct():
    rng = np.random.RandomState(0)
    M = (rng.randn(37, 256) * np.exp(3 * rng.randn(37, 1))).astype(np.float32)
    M[5] = 0.0                                            # an all-zero row must not divide by zero
    for nl in (2, 3):
        q, s = lp.limbs(M, 1, nl)
        assert s.shape == (37, 1) and np.all(s > 0)
        for i, qi in enumerate(q):
            assert np.all(qi == np.rint(qi))
            lo, hi = (-127, 127) if i == nl - 1 else (-64, 63)
            assert qi.min() >= lo and qi.max() <= hi, (nl, i, qi.min(), qi.max())
        full = sum(128.0 ** i * qi for i, qi in enumerate(q))
        np.testing.assert_array_equal(full, np.rint(M.astype(np.float64) / s))
        top = 127 * 128 ** (nl - 1) + sum(63 * 128 ** i for i in range(nl - 1))
        assert np.abs(full).max() <= top
        # the row maximum uses the whole range; the rounding error is half a step of the row's scale
        assert np.all(np.abs(full * s - M) <= 0.5 * s + 1e-30)


def test_three_term_product_is_the_four_term_one_minus_low_times_low():
    rng = np.random.RandomState(1)
    A = rng.randn(19, 256).astype(np.float32)
    B = (rng.rand(256, 23).astype(np.float32) - 0.5) / 8
    (a0, a1), sa = lp.limbs(A, 1, 2)
    (b0, b1), sb = lp.limbs(B, 0, 2)
    full = lp.mm(A, B, "i8x2f")
    np.testing.assert_allclose(full, ((128 * a1 + a0) * sa) @ ((128 * b1 + b0) * sb), rtol=0, atol=1e-12 * np.abs(full).max())
    np.testing.assert_allclose(lp.mm(A, B, "i8x2"), full - (a0 @ b0) * sa * sb, rtol=0, atol=1e-12 * np.abs(full).max())
    # i32 accumulator, in units of 128 (a0.b0 is dropped, so nothing sits below that): the a1.b1 sums are shifted left by
    # 7 once, then the two cross terms accumulate on top - |128 a1.b1 + a1.b0 + a0.b1| at K = 256 stays below 2^31
    worst = 256 * (128 * 127 * 127 + 2 * 127 * 64)
    assert worst < 2 ** 31
    # and the whole product is good to ~2^-15 of the row / column maxima
    err = np.abs(lp.mm(A, B, "i8x2") - A.astype(np.float64) @ B.astype(np.float64)).max()
    assert err < 256 ** 0.5 * 4 * 2.0 ** -15 * np.abs(A).max() * np.abs(B).max()


def test_bf16_helper_is_round_to_nearest_even():
    rng = np.random.RandomState(2)
    x = np.concatenate([rng.randn(4096).astype(np.float32), np.array([1.00390625, 1.01171875, -3.0, 0.0], np.float32)])
    ref = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    np.testing.assert_array_equal(lp.bf16_rne(x).astype(np.float32), ref)
    hi = lp.bf16_rne(x)
    lo = lp.bf16_rne(x.astype(np.float64) - hi)
    assert np.all(np.abs(hi + lo - x) <= 2.0 ** -16 * np.abs(x) + 1e-45)     # what bf16x3 keeps of an operand
