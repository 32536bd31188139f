"""BASELINE-size (6x256 net, 360 000 collocation points) checks on the GPU through
size-independent properties - the fp64 oracle cannot run the whole set in seconds:
  * pointwise: residuals of a random sample of points (incl. the last tile) equal the
    oracle's (points are independent);
  * shard additivity: gradients / loss sums of two half shards normalised by the GLOBAL
    count add up to the full-batch result (this is the multi-GPU contract);
  * permutation invariance of loss and gradient;
  * forward/backward consistency: a central finite difference of the forward loss along a
    random direction equals grad . direction.
Both precision modes run all four: fp32 (f32-input MFMA, 64-column tiles at hidden 256) and bf16x3 (the
128-column bf16 MFMA kernels that produce bench.py's headline number).  Tolerances per mode in TOL below
(bf16x3: residuals <= 2e-4 of max, the north star's 1e-4 on the loss).
"""
import numpy as np
import pytest
import torch

from oracle import autograd_ref as ar
from oracle import fwdmode_ref as fr

pytestmark = pytest.mark.gpu

L, H, RE, GRID = 6, 256, 2000.0, 600


# eq: residual max-abs / max|ref| ; u: abs ; sums: rel between two evaluations of the same points in another
# tile order ; grad: rel-L2 between them
TOL = {"fp32": dict(eq=2e-5, u=3e-6, sums=2e-6, grad=1e-5), "bf16x3": dict(eq=2e-4, u=1e-5, sums=5e-6, grad=3e-5)}


def _setup(x, y, n_global=None, flat=None, precision="fp32"):
    from nsfnet_amd import engine as eng
    E = eng.PinnEngine(torch.device("cuda:0"), L, H, RE, alpha_b=10.0, alpha_e=1.0, precision=precision)
    E.net.set_flat(torch.tensor(flat))
    E.set_collocation(x, y, n_global=n_global)
    xb, yb, ub, vb = (a.reshape(-1).astype(np.float32) for a in ar.cavity_boundary())
    E.set_boundary(xb, yb, ub, vb)
    return E


@pytest.fixture(scope="module", params=["fp32", "bf16x3"])
def base(request):
    prec = request.param
    flat = ar.flat_params(ar.seeded_net(3, L, H, seed=1234)).numpy().copy()
    x, y = (a.reshape(-1).astype(np.float32) for a in ar.uniform_grid(GRID, GRID))
    E = _setup(x, y, flat=flat, precision=prec)
    assert E.net.precision == (prec,) * 3
    E.loss_and_grad()
    torch.cuda.synchronize()
    yield dict(flat=flat, x=x, y=y, E=E, grads=E.grads.cpu().numpy().astype(np.float64), prec=prec, tol=TOL[prec],
               sums=E.sums.cpu().numpy().astype(np.float64), loss=float(E.loss_terms()["loss"]))
    del E
    torch.cuda.empty_cache()


def _rel_l2(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_sampled_points_match_oracle(base):
    rng = np.random.RandomState(0)
    n = base["x"].size
    idx = np.unique(np.concatenate([rng.randint(0, n, 300), np.arange(n - 40, n), np.arange(0, 40)]))
    P = fr.unflatten(base["flat"].astype(np.float64), 2, 3, L, H)
    out, _ = fr.forward4(P, base["x"][idx].astype(np.float64), base["y"][idx].astype(np.float64))
    eqs = fr.residuals(out, RE)
    f = base["E"].plan_f
    for k, name in enumerate(("eq1", "eq2", "eq3")):
        mine = f.field(name).cpu().numpy()[idx]
        assert np.abs(mine - eqs[k]).max() < base["tol"]["eq"] * np.abs(eqs[k]).max(), name
    np.testing.assert_allclose(f.field("u").cpu().numpy()[idx], out[:, 0, 0], atol=base["tol"]["u"])


def test_full_batch_loss_against_oracle_sample_mean(base):
    """The loss terms are means over points: the oracle's mean over a 4 000-point random sample of the grid
    must agree with the device's full 360 000-point mean within sampling error (3 sigma / sqrt(n)) - a
    whole-batch check that no tile is dropped, duplicated or mis-weighted at this size."""
    rng = np.random.RandomState(7)
    n = base["x"].size
    idx = rng.choice(n, 4000, replace=False)
    P = fr.unflatten(base["flat"].astype(np.float64), 2, 3, L, H)
    out, _ = fr.forward4(P, base["x"][idx].astype(np.float64), base["y"][idx].astype(np.float64))
    eqs = fr.residuals(out, RE)
    for k in range(3):
        sq = eqs[k] ** 2
        full = base["sums"][k] / n
        assert abs(sq.mean() - full) < 4.0 * sq.std() / np.sqrt(sq.size), (k, sq.mean(), full)


def test_shard_additivity(base):
    n = base["x"].size
    half = n // 2
    g, s = np.zeros_like(base["grads"]), np.zeros(4)
    for lo, hi in ((0, half), (half, n)):
        E = _setup(base["x"][lo:hi], base["y"][lo:hi], n_global=n, flat=base["flat"], precision=base["prec"])
        E.n_b_global = 2 * 2052            # each shard sees the whole BC set here: halve its weight
        E.loss_and_grad()
        g += E.grads.cpu().numpy().astype(np.float64)
        s += E.sums.cpu().numpy().astype(np.float64)[:4]
        del E
        torch.cuda.empty_cache()
    np.testing.assert_allclose(s[:3], base["sums"][:3], rtol=base["tol"]["sums"])
    assert _rel_l2(g, base["grads"]) < base["tol"]["grad"]


def test_permutation_invariance(base):
    perm = np.random.RandomState(1).permutation(base["x"].size)
    E = _setup(base["x"][perm], base["y"][perm], flat=base["flat"], precision=base["prec"])
    E.loss_and_grad()
    np.testing.assert_allclose(E.sums.cpu().numpy()[:3], base["sums"][:3], rtol=base["tol"]["sums"])
    assert _rel_l2(E.grads.cpu().numpy().astype(np.float64), base["grads"]) < base["tol"]["grad"]


def test_directional_derivative(base):
    rng = np.random.RandomState(2)
    d = rng.choice([-1.0, 1.0], size=base["flat"].size)
    eps = 2e-4
    E = base["E"]
    vals = []
    for sgn in (+1.0, -1.0):
        E.net.set_flat(torch.tensor((base["flat"].astype(np.float64) + sgn * eps * d).astype(np.float32)))
        E.loss_and_grad()
        s = E.sums.cpu().numpy().astype(np.float64)
        vals.append(10.0 * (s[8] + s[9]) / 2052 + (s[0] + s[1] + s[2]) / base["x"].size)
    E.net.set_flat(torch.tensor(base["flat"]))
    fd = (vals[0] - vals[1]) / (2 * eps)
    gd = float(base["grads"] @ d)
    assert abs(fd - gd) < 2e-2 * abs(gd), (fd, gd)
