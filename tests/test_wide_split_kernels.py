"""The role-split sweeps for WIDE nets (256 < hidden <= 448: fwd_bf16_wsplit.hip / bwd_bf16_wsplit.hip, BASELINE config 5's
8x400 among them) against the fp64 oracle and against the 8-wave wide kernels they replace ($PINN_WSPLIT=0), which
write and read the same S / Z-bar blocks: every padded width the geometry supports (9 .. 14 feature blocks: three or
four K regions, waves with and without a block in the last region), even / odd / single tile counts, ragged point
counts, 2 .. 8 hidden layers, both bf16 modes, plain and ev flavour; hidden 480 / 512 stay on the 8-wave kernels."""
import numpy as np
import pytest
import torch

from oracle import autograd_ref as ar
from oracle import fwdmode_ref as fr

pytestmark = pytest.mark.gpu


def _run(monkeypatch, wsplit, L, H, N, prec, ev=False, seed=0):
    from nsfnet_amd import engine as eng
    monkeypatch.setenv("PINN_WSPLIT", "1" if wsplit else "0")
    dev = torch.device("cuda:0")
    flat = ar.flat_params(ar.seeded_net(3, L, H, seed=70 + L + seed)).numpy().copy()
    rng = np.random.RandomState(N + H)
    x = rng.rand(N).astype(np.float32); y = rng.rand(N).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[::16].astype(np.float32) for a in ar.cavity_boundary())
    kw = dict(flavour="ev", n_hidden_e=2, hidden_e=24, alpha_evm=0.05) if ev else {}
    E = eng.PinnEngine(dev, L, H, 1500.0, alpha_b=10.0, alpha_e=1.0, precision=prec, **kw)
    E.net.set_flat(torch.tensor(flat))
    if ev:
        E.net_e.set_flat(ar.flat_params(ar.seeded_net(1, 2, 24, seed=3)))
        E.e_trainable = True
    E.set_collocation(x, y, weights=(0.5 + rng.rand(N)).astype(np.float32) if ev else None)
    E.set_boundary(xb, yb, ub, vb)
    names = E.plan_f.kernel_names()
    E.loss_and_grad()
    torch.cuda.synchronize()
    out = dict(fields=E.plan_f.fields[:, :N].cpu().numpy().astype(np.float64), sums=E.sums.cpu().numpy().astype(np.float64),
               grads=E.grads.cpu().numpy().astype(np.float64), flat=flat, x=x, y=y, names=names,
               bc=(xb, yb, ub, vb))
    if ev:
        out["grads_e"] = E.grads_e.cpu().numpy().astype(np.float64)
    return out


def _rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("H,L,N", [(400, 8, 330), (400, 8, 16), (288, 3, 97), (320, 2, 640), (330, 4, 49), (384, 5, 200),
                                   (448, 3, 171), (416, 6, 1)])
@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_wide_split_matches_the_8wave_wide_kernels_and_the_oracle(monkeypatch, H, L, N, prec):
    a = _run(monkeypatch, True, L, H, N, prec)
    b = _run(monkeypatch, False, L, H, N, prec)
    assert a["names"][:2] == ("fwd_wsplit_kernel", "bwd_wsplit_kernel") and a["names"][2] == "dw_bf16_wide_kernel"
    assert b["names"][:2] == ("fwd_bf16_wide_kernel", "bwd_bf16_wide_kernel")
    # the two schedules sum the bias and the K range in a different order: they agree to the rounding of the products
    tol = 3e-4 if prec == "bf16x3" else 5e-2
    allmax = np.abs(b["fields"]).max()
    for k in range(a["fields"].shape[0]):
        # (a plane's own maximum, but not below 1 % of the largest plane: with a handful of points a derivative plane can
        # be a cancellation to 1e-5 of the values it is made of)
        scale = max(np.abs(b["fields"][k]).max(), 1e-2 * allmax, 1e-30)
        assert np.abs(a["fields"][k] - b["fields"][k]).max() <= tol * scale, k
    np.testing.assert_allclose(a["sums"][:4], b["sums"][:4], rtol=tol, atol=1e-30)
    assert _rel_l2(a["grads"], b["grads"]) < (1e-4 if prec == "bf16x3" else 5e-2)
    if prec == "bf16x3":      # the parity statement: the fp64 oracle at the bf16x3 bars
        P = fr.unflatten(a["flat"].astype(np.float64), 2, 3, L, H)
        r = fr.pde_loss_and_grad(P, a["x"].astype(np.float64), a["y"].astype(np.float64), 1500.0, alpha_e=1.0)
        bb = fr.bc_loss_and_grad(P, *(v.astype(np.float64) for v in a["bc"]), alpha_b=10.0)
        for k, name in enumerate(("eq1", "eq2", "eq3")):
            err = np.abs(a["fields"][6 + k] - r["eqs"][k]).max() / max(np.abs(r["eqs"][k]).max(), 1e-30)
            assert err < 5e-4, (name, err)
        np.testing.assert_allclose(a["sums"][0:3], r["sums"], rtol=2e-4)
        assert _rel_l2(a["grads"], r["grad"] + bb["grad"]) < 1e-4


def test_wide_split_ev_flavour(monkeypatch):
    """ev-NSFnet terms (eq4, lagged viscosity, per-point weights, d loss / d e into the entropy net) through the wide
    role-split sweeps: BASELINE config 5's flavour (ev-NSFnet/pinn_solver.py:290-342)."""
    a = _run(monkeypatch, True, 4, 400, 210, "bf16x3", ev=True)
    b = _run(monkeypatch, False, 4, 400, 210, "bf16x3", ev=True)
    assert a["names"][0] == "fwd_wsplit_kernel"
    for k in range(a["fields"].shape[0]):
        scale = max(np.abs(b["fields"][k]).max(), 1e-30)
        assert np.abs(a["fields"][k] - b["fields"][k]).max() <= 3e-4 * scale, k
    assert _rel_l2(a["grads"], b["grads"]) < 1e-4
    assert _rel_l2(a["grads_e"], b["grads_e"]) < 1e-4


@pytest.mark.parametrize("H", [480, 512])
def test_widths_whose_last_region_does_not_fit_twice_keep_the_8wave_kernels(monkeypatch, H):
    a = _run(monkeypatch, True, 2, H, 64, "bf16x3")
    assert a["names"][:2] == ("fwd_bf16_wide_kernel", "bwd_bf16_wide_kernel")
