"""CPU model of the index arithmetic of the wide role-split sweeps (nsfnet_amd/csrc/fwd_bf16_wsplit.hip, bwd_bf16_wsplit.hip):
which (wave, block, quad, lane row, element) owns which feature, where a feature's 8-element chunk sits in the shared image
for each group (the last K region has one copy per group), and the order in which the reverse sweep walks a phase's
register quads.  The formulas are restated here and the source is checked to still contain them, so a changed constant
fails on the CPU before it reaches a GPU (the end-to-end statement is tests/test_wide_split_kernels.py)."""
import os

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nsfnet_amd", "csrc")
SUPPORTED = (288, 320, 352, 384, 416, 448)


def geo(HP):
    NB = HP // 32
    MQ = (NB + 3) // 4
    LASTK = 128 * (MQ - 1)
    return dict(NB=NB, MQ=MQ, KS=HP // 16, LASTK=LASTK, LASTN=HP - LASTK, RSE=512)


def test_source_still_states_the_modelled_formulas():
    f = open(os.path.join(CSRC, "fwd_bf16_wsplit.hip")).read()
    b = open(os.path.join(CSRC, "bwd_bf16_wsplit.hip")).read()
    for src in (f, b):
        assert "NB = HP / 32, MQ = (NB + 3) / 4, KS = HP / 16" in src
        assert "(NB - w + 3) / 4" in src
        assert "(o >> 3) + ((o >= G::LASTK && g) ? G::LASTN / 8 : 0)" in src
        assert "XI::RSE - HP >= LASTN" in src
    assert "32 * (4 * bq + w) + 8 * (k + 2 * hi) + 4 * h" in f and "32 * (4 * bq + w) + 8 * (k + 2 * hi) + 4 * h" in b
    assert "i < 6 ? ((i % 3) == 2 ? MQ - 1 : i / 3) : 2 + (i - 6) / 2" in b
    assert "i < 6 ? ((i % 3) == 2 ? i / 3 : i % 3) : (i - 6) % 2" in b


@pytest.mark.parametrize("HP", SUPPORTED)
def test_every_feature_has_exactly_one_owner(HP):
    g = geo(HP)
    seen = {}
    for w in range(4):
        mc = (g["NB"] - w + 3) // 4
        assert g["MQ"] - 1 <= mc <= g["MQ"]              # only the LAST block of a wave can be missing
        for bq in range(mc):
            for k in range(2):
                for hi in range(2):
                    for h in range(2):
                        for e in range(4):
                            o = 32 * (4 * bq + w) + 8 * (k + 2 * hi) + 4 * h + e
                            assert o not in seen, (o, seen.get(o))
                            seen[o] = (w, bq, k, hi, h, e)
    assert sorted(seen) == list(range(HP))
    # region q of the image (128 features) holds block q of every wave: what the M group reads in quarter q is what the
    # E group computed as its block q
    for o, (w, bq, *_r) in seen.items():
        assert o // 128 == bq


@pytest.mark.parametrize("HP", SUPPORTED + (480, 512))
def test_last_region_copies_fit_behind_the_features_or_the_width_is_excluded(HP):
    g = geo(HP)
    fits = g["RSE"] - HP >= g["LASTN"]
    assert fits == (HP in SUPPORTED)
    if not fits:
        return
    chunks = {0: set(), 1: set()}
    for grp in (0, 1):
        for o in range(0, HP, 8):
            c = (o >> 3) + (g["LASTN"] // 8 if (o >= g["LASTK"] and grp) else 0)
            assert 0 <= c < g["RSE"] // 8
            chunks[grp].add(c)
    shared = set(range(g["LASTK"] // 8))
    assert chunks[0] & chunks[1] == shared                # the groups share regions 0 .. MQ-2 and nothing else
    assert len(chunks[0]) == len(chunks[1]) == HP // 8
    # the k-step -> chunk map of the M phase (chunk of feature 16 s, + h) agrees with the E phase's dump position
    for grp in (0, 1):
        for s in range(g["KS"]):
            for h in range(2):
                c = ((16 * s) >> 3) + (g["LASTN"] // 8 if (16 * s >= g["LASTK"] and grp) else 0) + h
                o = 16 * s + 8 * h
                assert c == (o >> 3) + (g["LASTN"] // 8 if (o >= g["LASTK"] and grp) else 0)


@pytest.mark.parametrize("MQ", [3, 4])
def test_reverse_sweep_item_order_covers_every_quad_once(MQ):
    """Quarter 0: (0,0) (0,1) (MQ-1,0); quarter 1: (1,0) (1,1) (MQ-1,1); quarter q >= 2: (q,0) (q,1); last quarter: none -
    2 MQ items, each (block, quad) once, and an item's saved-activation slot (i mod 3) is free when its load is issued
    two items ahead."""
    item_bq = lambda i: (MQ - 1 if i % 3 == 2 else i // 3) if i < 6 else 2 + (i - 6) // 2
    item_k = lambda i: (i // 3 if i % 3 == 2 else i % 3) if i < 6 else (i - 6) % 2
    items = [(item_bq(i), item_k(i)) for i in range(2 * MQ)]
    assert sorted(items) == [(b, k) for b in range(MQ) for k in range(2)]
    assert items[:6] == [(0, 0), (0, 1), (MQ - 1, 0), (1, 0), (1, 1), (MQ - 1, 1)]
    for i in range(2 * MQ - 2):
        assert (i + 2) % 3 not in (i % 3, (i + 1) % 3)     # the ring slot being refilled is neither in use nor next
