"""The 24-bit spill format of the role-split / wide bf16 sweeps (nsfnet_amd/csrc/bf16_util.h pack24 / unpack24), checked on
the CPU: the v_perm_b32 byte selectors are read from the header and emulated, so a changed constant fails here before it
reaches a GPU.  (The end-to-end statement - the kernels with this format against the fp64 oracle - is the GPU suite.)"""
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nsfnet_amd", "csrc", "bf16_util.h")


def v_perm_b32(s0, s1, sel):
    """D.byte[i] = selector byte i: 0-3 -> byte of s1, 4-7 -> byte of s0, 0x0c -> 0x00 (gfx9 ISA)."""
    out = 0
    for i in range(4):
        k = (sel >> (8 * i)) & 0xff
        if k <= 3:
            b = (s1 >> (8 * k)) & 0xff
        elif k <= 7:
            b = (s0 >> (8 * (k - 4))) & 0xff
        else:
            assert k == 0x0c, hex(k)
            b = 0
        out |= b << (8 * i)
    return out


def _selectors():
    src = open(HDR).read()
    body = src[src.index("void pack24("):src.index("// tanh for the bf16 modes")]
    sels = [int(m, 16) for m in re.findall(r"0x([0-9a-fA-F]{8})u", body)]
    # pack24: hi pair selector (twice), two lo selectors; unpack24: four selectors
    assert len(sels) >= 8, sels
    return sels[:8]


def test_pack24_round_trip_is_round_to_24_bits():
    hi_a, hi_b, lo_a, lo_b, u0, u1, u2, u3 = _selectors()
    assert hi_a == hi_b
    rng = np.random.RandomState(0)
    x = (rng.randn(2000) * 10.0 ** rng.uniform(-8, 4, 2000)).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 1.0000001, 3.4e38, 1e-38, -2.5]
    worst = 0.0
    for q in x.reshape(-1, 4):
        r = [(int(v) + 0x80) & 0xffffffff for v in q.view(np.uint32)]
        hi0, hi1 = v_perm_b32(r[1], r[0], hi_a), v_perm_b32(r[3], r[2], hi_b)
        lo = v_perm_b32(r[1], r[0], lo_a) | v_perm_b32(r[3], r[2], lo_b)
        y = np.array([v_perm_b32(hi0, lo, u0), v_perm_b32(hi0, lo, u1), v_perm_b32(hi1, lo, u2), v_perm_b32(hi1, lo, u3)],
                     dtype=np.uint32)
        for a, b, rr in zip(q, y.view(np.float32), r):
            assert np.uint32(rr & 0xffffff00) == np.float32(b).view(np.uint32)      # = x rounded half-up at bit 8
            if np.isfinite(b) and a != 0:
                worst = max(worst, abs(float(b) - float(a)) / abs(float(a)))
        # the packed planes are what the hardware sees: bf16 top halves and the third bytes
        assert hi0 == ((r[0] >> 16) | (r[1] & 0xffff0000)) and hi1 == ((r[2] >> 16) | (r[3] & 0xffff0000))
        assert lo == sum(((r[i] >> 8) & 0xff) << (8 * i) for i in range(4))
    assert worst <= 2.0 ** -16


def test_pack24_keeps_the_nans_and_infinities_arithmetic_produces():
    """A diverged activation must still poison the reverse sweep after the spill.  Pinned: +-infinity, the hardware's
    quiet NaNs and every quiet NaN with a payload below 0x7fff80 keep their class; the largest finite value rounds up to
    infinity.  Documented in bf16_util.h and NOT guarded (three VALU per value in the hottest loop): an all-ones payload
    carries into the exponent and reads back as zero - no instruction of the sweeps produces such a NaN."""
    hi_a, hi_b, lo_a, lo_b, u0, u1, u2, u3 = _selectors()
    cases = [0x7fc00000, 0xffc00000, 0x7fc00001, 0xffd12345, 0x7fff7f7f, 0x7f800000, 0xff800000, 0x7f7fffff]
    for k in range(0, len(cases), 4):
        r = [(u + 0x80) & 0xffffffff for u in cases[k:k + 4]]
        hi0, hi1 = v_perm_b32(r[1], r[0], hi_a), v_perm_b32(r[3], r[2], hi_b)
        lo = v_perm_b32(r[1], r[0], lo_a) | v_perm_b32(r[3], r[2], lo_b)
        y = np.array([v_perm_b32(hi0, lo, u0), v_perm_b32(hi0, lo, u1), v_perm_b32(hi1, lo, u2), v_perm_b32(hi1, lo, u3)],
                     dtype=np.uint32).view(np.float32)
        for u, b in zip(cases[k:k + 4], y):
            a = np.array([u], dtype=np.uint32).view(np.float32)[0]
            if np.isnan(a):
                assert np.isnan(b), hex(u)
            elif np.isinf(a):
                assert b == a, hex(u)
            else:
                assert np.isinf(b) and np.sign(b) == np.sign(a), hex(u)      # the largest finite value rounds up to infinity
    # the documented hole: the carry of an all-ones payload
    assert ((0x7fffffff + 0x80) & 0xffffff00) == 0x80000000
