"""The software-pipelined (one wave per SIMD, two tiles in flight) bf16 forward and reverse sweeps against the 8-wave kernels they
replace for hidden = 256 and against the fp64 oracle: same fields, loss sums, saved activations (seen through the
gradient the reverse sweep computes from them) for even / odd / single tile counts, ragged point counts, 2..7 hidden
layers, both bf16 modes, plain and ev flavour.  The two schedules sum in a different order (bias added after the
GEMM instead of seeding it), so they agree to the rounding of the bf16x3 products (3e-5 of the field max), not bit for bit; the fp64 oracle check at the
bf16x3 bars is the parity statement."""
import numpy as np
import pytest
import torch

from oracle import autograd_ref as ar
from oracle import fwdmode_ref as fr

pytestmark = pytest.mark.gpu
H = 256


def _run(monkeypatch, pipe, L, N, prec, ev=False):
    """pipe: False = 8-wave kernels, True = one-wave-per-SIMD pipelined sweeps, "split" = role-split forward (two
    wave groups in opposite phases) + role-split reverse sweep."""
    from nsfnet_amd import engine as eng
    monkeypatch.setenv("PINN_SCHED", {False: "0", True: "1", "split": "2"}[pipe])
    for k in ("PINN_FWD_SCHED", "PINN_BWD_SCHED"):
        monkeypatch.delenv(k, raising=False)
    dev = torch.device("cuda:0")
    flat = ar.flat_params(ar.seeded_net(3, L, H, seed=40 + L)).numpy().copy()
    rng = np.random.RandomState(N)
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
    E.loss_and_grad()
    torch.cuda.synchronize()
    out = dict(fields=E.plan_f.fields[:, :N].cpu().numpy().astype(np.float64), sums=E.sums.cpu().numpy().astype(np.float64),
               grads=E.grads.cpu().numpy().astype(np.float64), flat=flat, x=x, y=y,
               vis=E.plan_f.vis_t.cpu().numpy().copy())
    if ev:
        out["grads_e"] = E.grads_e.cpu().numpy().astype(np.float64)
    return out


@pytest.mark.parametrize("L,N", [(6, 320), (6, 330), (6, 20), (2, 97), (3, 640), (7, 65), (4, 2049)])
@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("sched", [True, "split"])
def test_pipelined_forward_matches_8wave_kernels(monkeypatch, L, N, prec, sched):
    a = _run(monkeypatch, sched, L, N, prec)
    b = _run(monkeypatch, False, L, N, prec)
    # bf16x3: the hi/lo splits of slightly different intermediates round differently (~2^-17 per product, more on the
    # small derivative planes); plain bf16: operands rounded to 8 bits, order effects are plainly visible
    tol = 2e-4 if prec == "bf16x3" else 5e-2
    for k in range(a["fields"].shape[0]):
        scale = max(np.abs(b["fields"][k]).max(), 1e-30)
        assert np.abs(a["fields"][k] - b["fields"][k]).max() <= tol * scale, k
    np.testing.assert_allclose(a["sums"][:4], b["sums"][:4], rtol=tol, atol=1e-30)
    assert np.linalg.norm(a["grads"] - b["grads"]) <= (1e-4 if prec == "bf16x3" else 5e-2) * np.linalg.norm(b["grads"])
    if prec == "bf16x3":      # and against the fp64 oracle at the bf16x3 bars
        P = fr.unflatten(a["flat"].astype(np.float64), 2, 3, L, H)
        r = fr.pde_loss_and_grad(P, a["x"].astype(np.float64), a["y"].astype(np.float64), 1500.0, alpha_e=1.0)
        for k, name in ((6, "eq1"), (7, "eq2"), (8, "eq3")):
            assert np.abs(a["fields"][k] - r["eqs"][k - 6]).max() < 5e-4 * np.abs(r["eqs"][k - 6]).max(), name
        np.testing.assert_allclose(a["sums"][:3], r["sums"], rtol=2e-4)
        xb, yb, ub, vb = (q.reshape(-1)[::16].astype(np.float64) for q in ar.cavity_boundary())
        bq = fr.bc_loss_and_grad(P, xb.astype(np.float32).astype(np.float64), yb.astype(np.float32).astype(np.float64),
                                 ub.astype(np.float32), vb.astype(np.float32), alpha_b=10.0)
        g_ref = r["grad"] + bq["grad"]
        assert np.linalg.norm(a["grads"] - g_ref) <= 1e-4 * np.linalg.norm(g_ref)      # pipelined reverse sweep vs fp64


@pytest.mark.parametrize("sched", [True, "split"])
def test_pipelined_forward_ev_flavour(monkeypatch, sched):
    a = _run(monkeypatch, sched, 5, 450, "bf16x3", ev=True)
    b = _run(monkeypatch, False, 5, 450, "bf16x3", ev=True)
    for k in range(a["fields"].shape[0]):
        scale = max(np.abs(b["fields"][k]).max(), 1e-30)
        assert np.abs(a["fields"][k] - b["fields"][k]).max() <= 2e-4 * scale, k
    np.testing.assert_array_equal(a["vis"], b["vis"])
    np.testing.assert_allclose(a["sums"][:4], b["sums"][:4], rtol=1e-4)
    assert np.linalg.norm(a["grads"] - b["grads"]) <= 1e-4 * np.linalg.norm(b["grads"])
    assert np.linalg.norm(a["grads_e"] - b["grads_e"]) <= 1e-4 * np.linalg.norm(b["grads_e"])
