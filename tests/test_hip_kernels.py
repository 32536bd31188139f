"""GPU parity of the HIP pipeline (through the C ABI) against the fp64 forward-mode
oracle on seeded inputs.  Tolerances (fp32 arithmetic):
  fields / residuals  max-abs error <= 2e-5 * max|ref|   (noise floor ~2-6e-6)
  loss sums           relative <= 1e-5
  gradients           relative L2 <= 1e-4 (measured ~1e-6)
"""
import numpy as np
import pytest
import torch

from oracle import autograd_ref as ar
from oracle import fwdmode_ref as fr

pytestmark = pytest.mark.gpu


def _engine_mod():
    from nsfnet_amd import engine
    return engine


def _rel_max(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _rand_params(n_out, L, H, seed):
    net = ar.seeded_net(n_out, L, H, seed=seed)
    return ar.flat_params(net).numpy().copy()


CASES = [  # (L, H, N, Nb)   H=50 / 40 exercise the zero padding to 64, N not a tile multiple
    (1, 8, 37, 5), (2, 16, 300, 33), (4, 50, 1000, 129), (3, 96, 257, 64), (6, 128, 520, 200), (6, 256, 320, 100),
    (3, 200, 96, 40),
    # wide nets (64-column tiles, permlane16 stream exchange, blocked dW): BASELINE config 5 is 8x400
    (2, 300, 70, 30), (3, 400, 150, 70), (2, 512, 40, 64),
]


@pytest.mark.parametrize("L,H,N,Nb", CASES)
def test_residual_and_bc_gradients(L, H, N, Nb):
    eng = _engine_mod()
    dev = torch.device("cuda:0")
    Re, alpha_b, alpha_e = 400.0, 10.0, 1.0
    flat = _rand_params(3, L, H, seed=100 + H)
    rng = np.random.RandomState(H + L)
    x = rng.rand(N).astype(np.float32); y = rng.rand(N).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[:: max(1, 2052 // Nb)][:Nb].astype(np.float32) for a in ar.cavity_boundary())
    E = eng.PinnEngine(dev, L, H, Re, alpha_b=alpha_b, alpha_e=alpha_e)
    E.net.set_flat(torch.tensor(flat))
    E.set_collocation(x, y)
    E.set_boundary(xb, yb, ub, vb)
    E.loss_and_grad()
    torch.cuda.synchronize()
    P = fr.unflatten(flat.astype(np.float64), 2, 3, L, H)
    r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), Re, alpha_e=alpha_e)
    b = fr.bc_loss_and_grad(P, xb.astype(np.float64), yb.astype(np.float64), ub, vb, alpha_b=alpha_b)
    out = r["out"]
    f = E.plan_f
    for name, ref in (("u", out[:, 0, 0]), ("v", out[:, 1, 0]), ("p", out[:, 2, 0]), ("u_x", out[:, 0, 1]),
                      ("u_y", out[:, 0, 2]), ("v_x", out[:, 1, 1]), ("v_y", out[:, 1, 2]),
                      ("eq1", r["eqs"][0]), ("eq2", r["eqs"][1]), ("eq3", r["eqs"][2])):
        assert _rel_max(f.field(name).cpu().numpy(), ref) < 2e-5, name
    sums = E.sums.cpu().numpy()
    np.testing.assert_allclose(sums[0:3], r["sums"], rtol=1e-5)
    np.testing.assert_allclose(sums[8:10], b["sums"], rtol=1e-5)
    np.testing.assert_allclose(E.plan_b.pred.cpu().numpy()[:2].T, b["pred"][:, :2], rtol=0, atol=2e-6)
    g = E.grads.cpu().numpy()
    assert _rel_l2(g, r["grad"] + b["grad"]) < 1e-4
    lt = E.loss_terms()
    ref_loss = alpha_b * sum(b["sums"]) / Nb + alpha_e * sum(r["sums"]) / N
    assert abs(float(lt["loss"]) - ref_loss) < 1e-5 * ref_loss


def test_ev_residuals_weights_scale_and_entropy_gradient():
    eng = _engine_mod()
    dev = torch.device("cuda:0")
    L, H, L1, H1, N, Re = 4, 50, 4, 40, 777, 4000.0
    flat = _rand_params(3, L, H, seed=5); flat_e = _rand_params(1, L1, H1, seed=6)
    rng = np.random.RandomState(3)
    x = (rng.rand(N) * 2 - 1).astype(np.float32); y = (rng.rand(N) * 2 - 1).astype(np.float32)
    w = (0.3 + rng.rand(N)).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[::16].astype(np.float32) for a in ar.cavity_boundary())
    E = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0, flavour="ev", n_hidden_e=L1, hidden_e=H1,
                       alpha_evm=0.05, coord_scale=2.0)
    E.net.set_flat(torch.tensor(flat)); E.net_e.set_flat(torch.tensor(flat_e))
    E.e_trainable = True
    E.set_collocation(x, y, weights=w)
    E.set_boundary(xb, yb, ub, vb)
    vtm0 = E.plan_f.vis_t_minus.cpu().numpy().astype(np.float64)
    E.loss_and_grad()
    torch.cuda.synchronize()
    P = fr.unflatten(flat.astype(np.float64), 2, 3, L, H)
    Pe = fr.unflatten(flat_e.astype(np.float64), 2, 1, L1, H1)
    e, saved_e = fr.forward1(Pe, x.astype(np.float64), y.astype(np.float64))
    np.testing.assert_allclose(vtm0, 0.05 * np.abs(e[:, 0]), rtol=1e-4, atol=1e-8)
    vis_t = np.minimum(np.float32(20.0 / Re), vtm0)
    np.testing.assert_allclose(E.plan_f.vis_t.cpu().numpy(), vis_t, rtol=1e-6)
    np.testing.assert_allclose(E.plan_f.vis_t_minus.cpu().numpy(), 0.05 * np.abs(e[:, 0]), rtol=1e-4, atol=1e-8)
    r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), Re, alpha_e=1.0, vis_t=vis_t,
                             e=e[:, 0], w=w.astype(np.float64), scale=2.0)
    b = fr.bc_loss_and_grad(P, xb.astype(np.float64), yb.astype(np.float64), ub, vb, alpha_b=10.0)
    for k, name in enumerate(("eq1", "eq2", "eq3", "eq4")):
        assert _rel_max(E.plan_f.field(name).cpu().numpy(), r["eqs"][k]) < 2e-5, name
    np.testing.assert_allclose(E.sums.cpu().numpy()[0:4], r["sums"], rtol=1e-5)
    assert _rel_l2(E.grads.cpu().numpy(), r["grad"] + b["grad"]) < 1e-4
    ge = fr.backward1(Pe, x.astype(np.float64), y.astype(np.float64), saved_e, r["e_adj"].reshape(-1, 1))
    assert _rel_l2(E.grads_e.cpu().numpy(), ge) < 1e-4


def test_adam_matches_torch():
    eng = _engine_mod()
    dev = torch.device("cuda:0")
    net = eng.DeviceNet(3, 2, 16, dev)
    torch.manual_seed(0)
    p0 = torch.randn(net.num_params)
    net.set_flat(p0)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    for k in range(5):
        g = torch.randn(net.num_params) * (10.0 ** (-k))
        ref.grad = g.clone()
        opt.step()
        net.adam_step(g.to(dev), 1e-3)
        np.testing.assert_allclose(net.params.cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=1e-8)


def test_predict_matches_forward1():
    eng = _engine_mod()
    dev = torch.device("cuda:0")
    L, H = 6, 256
    flat = _rand_params(3, L, H, seed=1234)
    E = eng.PinnEngine(dev, L, H, 2000.0)
    E.net.set_flat(torch.tensor(flat))
    x, y = ar.uniform_grid(33, 21)
    u, v, p = E.predict(x.astype(np.float32), y.astype(np.float32))
    out, _ = fr.forward1(fr.unflatten(flat.astype(np.float64), 2, 3, L, H), x.astype(np.float32), y.astype(np.float32))
    for mine, ref in ((u, out[:, 0]), (v, out[:, 1]), (p, out[:, 2])):
        np.testing.assert_allclose(mine.cpu().numpy(), ref, rtol=0, atol=3e-6)


# ---------------------------------------------------------------------------
# bf16x3 precision mode (3 bf16 MFMAs per product, fp32 accumulate).  North-star bar:
# per-step loss within 1e-4 relative of the reference; measured ~4e-7 (loss), <=1.3e-4
# (pointwise residuals, relative to max|eq|), ~3e-6 (gradient rel-L2).
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("L,H,N,Nb", [(2, 16, 300, 33), (4, 50, 1000, 129), (3, 96, 257, 64), (6, 128, 520, 200),
                                      (6, 256, 320, 100), (3, 200, 96, 40)])
@pytest.mark.parametrize("prec", ["bf16x3", ("bf16x3", "fp32", "fp32"), ("fp32", "bf16x3", "bf16x3")])
def test_bf16x3_mode_meets_parity_bar(L, H, N, Nb, prec):
    eng = _engine_mod()
    dev = torch.device("cuda:0")
    Re, alpha_b, alpha_e = 400.0, 10.0, 1.0
    flat = _rand_params(3, L, H, seed=100 + H)
    rng = np.random.RandomState(H + L)
    x = rng.rand(N).astype(np.float32); y = rng.rand(N).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[:: max(1, 2052 // Nb)][:Nb].astype(np.float32) for a in ar.cavity_boundary())
    E = eng.PinnEngine(dev, L, H, Re, alpha_b=alpha_b, alpha_e=alpha_e, precision=prec)
    E.net.set_flat(torch.tensor(flat))
    E.set_collocation(x, y)
    E.set_boundary(xb, yb, ub, vb)
    E.loss_and_grad()
    torch.cuda.synchronize()
    P = fr.unflatten(flat.astype(np.float64), 2, 3, L, H)
    r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), Re, alpha_e=alpha_e)
    b = fr.bc_loss_and_grad(P, xb.astype(np.float64), yb.astype(np.float64), ub, vb, alpha_b=alpha_b)
    for k, name in enumerate(("eq1", "eq2", "eq3")):
        assert _rel_max(E.plan_f.field(name).cpu().numpy(), r["eqs"][k]) < 5e-4, name
    np.testing.assert_allclose(E.sums.cpu().numpy()[0:3], r["sums"], rtol=2e-4)
    ref_loss = alpha_b * sum(b["sums"]) / Nb + alpha_e * sum(r["sums"]) / N
    assert abs(float(E.loss_terms()["loss"]) - ref_loss) < 1e-4 * ref_loss
    assert _rel_l2(E.grads.cpu().numpy(), r["grad"] + b["grad"]) < 1e-4


def test_bf16_fast_mode_runs_and_is_close():
    """Plain bf16 operands: reported-only fast mode (does NOT meet the 1e-4 bar; sanity bound 5 %)."""
    eng = _engine_mod()
    L, H, N = 6, 256, 640
    flat = _rand_params(3, L, H, seed=9)
    rng = np.random.RandomState(1)
    x = rng.rand(N).astype(np.float32); y = rng.rand(N).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[::16].astype(np.float32) for a in ar.cavity_boundary())
    out = {}
    for prec in ("fp32", "bf16"):
        E = eng.PinnEngine(torch.device("cuda:0"), L, H, 2000.0, alpha_b=10.0, alpha_e=1.0, precision=prec)
        E.net.set_flat(torch.tensor(flat)); E.set_collocation(x, y); E.set_boundary(xb, yb, ub, vb)
        E.loss_and_grad()
        out[prec] = (float(E.loss_terms()["loss"]), E.grads.cpu().numpy().astype(np.float64))
    assert abs(out["bf16"][0] - out["fp32"][0]) < 5e-2 * out["fp32"][0]
    assert _rel_l2(out["bf16"][1], out["fp32"][1]) < 5e-2


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_config5_shape_ev_8x400_with_4x40_entropy_net(prec):
    """BASELINE config 5 shape (ev-NSFnet Re=10000, 8x400 main net + 4x40 entropy net) at a
    test-sized point count, entropy net trainable, vs the fp64 oracle - in the bit-exact fp32 mode and in
    bf16x3 (the mode config 5 names: wide bf16 kernels for the main net, narrow ones for the entropy net)."""
    eng = _engine_mod()
    dev = torch.device("cuda:0")
    L, H, L1, H1, N, Re = 8, 400, 4, 40, 200, 10000.0
    flat = _rand_params(3, L, H, seed=15); flat_e = _rand_params(1, L1, H1, seed=16)
    rng = np.random.RandomState(8)
    x = rng.rand(N).astype(np.float32); y = rng.rand(N).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[::32].astype(np.float32) for a in ar.cavity_boundary())
    E = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0, flavour="ev", n_hidden_e=L1, hidden_e=H1,
                       alpha_evm=0.05, precision=prec)
    tol_eq, tol_sum = (5e-5, 2e-5) if prec == "fp32" else (5e-4, 2e-4)
    E.net.set_flat(torch.tensor(flat)); E.net_e.set_flat(torch.tensor(flat_e))
    E.e_trainable = True
    E.set_collocation(x, y)
    E.set_boundary(xb, yb, ub, vb)
    vtm0 = E.plan_f.vis_t_minus.cpu().numpy().astype(np.float64)
    E.loss_and_grad()
    torch.cuda.synchronize()
    P = fr.unflatten(flat.astype(np.float64), 2, 3, L, H)
    Pe = fr.unflatten(flat_e.astype(np.float64), 2, 1, L1, H1)
    e, saved_e = fr.forward1(Pe, x.astype(np.float64), y.astype(np.float64))
    vis_t = np.minimum(np.float32(20.0 / Re), vtm0)
    r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), Re, vis_t=vis_t, e=e[:, 0])
    b = fr.bc_loss_and_grad(P, xb.astype(np.float64), yb.astype(np.float64), ub, vb, alpha_b=10.0)
    for k, name in enumerate(("eq1", "eq2", "eq3", "eq4")):
        assert _rel_max(E.plan_f.field(name).cpu().numpy(), r["eqs"][k]) < tol_eq, name
    np.testing.assert_allclose(E.sums.cpu().numpy()[0:4], r["sums"], rtol=tol_sum)
    assert _rel_l2(E.grads.cpu().numpy(), r["grad"] + b["grad"]) < 1e-4
    ge = fr.backward1(Pe, x.astype(np.float64), y.astype(np.float64), saved_e, r["e_adj"].reshape(-1, 1))
    assert _rel_l2(E.grads_e.cpu().numpy(), ge) < 1e-4


@pytest.mark.parametrize("H,L", [(400, 8), (288, 3), (512, 2), (330, 4)])
def test_wide_nets_bf16x3_sweeps(H, L):
    """hidden > 256 in bf16x3 mode: fwd_bf16_wide / bwd_bf16_wide (two 32-feature blocks per wave; odd block
    counts: 288 = 9, 330 -> 352 = 11) + the fp32 dW kernel, against the fp64 oracle at the bf16x3 bar."""
    eng = _engine_mod()
    dev = torch.device("cuda:0")
    N, Re = 300, 2000.0
    flat = _rand_params(3, L, H, seed=21)
    rng = np.random.RandomState(9)
    x = rng.rand(N).astype(np.float32); y = rng.rand(N).astype(np.float32)
    w = (0.5 + rng.rand(N)).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[::16].astype(np.float32) for a in ar.cavity_boundary())
    E = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0, precision="bf16x3")
    E.net.set_flat(torch.tensor(flat))
    E.set_collocation(x, y, weights=w)
    E.set_boundary(xb, yb, ub, vb)
    E.loss_and_grad()
    torch.cuda.synchronize()
    P = fr.unflatten(flat.astype(np.float64), 2, 3, L, H)
    r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), Re, w=w.astype(np.float64))
    b = fr.bc_loss_and_grad(P, xb.astype(np.float64), yb.astype(np.float64), ub, vb, alpha_b=10.0)
    for k, name in enumerate(("eq1", "eq2", "eq3")):
        assert _rel_max(E.plan_f.field(name).cpu().numpy(), r["eqs"][k]) < 5e-4, name
    np.testing.assert_allclose(E.sums.cpu().numpy()[0:3], r["sums"], rtol=2e-4)
    np.testing.assert_allclose(E.sums.cpu().numpy()[8:10], b["sums"], rtol=2e-4)
    assert _rel_l2(E.grads.cpu().numpy(), r["grad"] + b["grad"]) < 1e-4
    lt = E.loss_terms()
    ref_loss = 10.0 * sum(b["sums"]) / len(xb) + sum(r["sums"]) / N
    assert abs(float(lt["loss"]) - ref_loss) < 1e-4 * ref_loss


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_wide_net_predict_and_fast_mode(prec):
    """8x400: value-mode prediction (evaluate / test path) and the plain-bf16 fast mode on the wide kernels."""
    eng = _engine_mod()
    dev = torch.device("cuda:0")
    L, H, N = 8, 400, 500
    flat = _rand_params(3, L, H, seed=31)
    rng = np.random.RandomState(10)
    x = rng.rand(N).astype(np.float32); y = rng.rand(N).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[::16].astype(np.float32) for a in ar.cavity_boundary())
    E = eng.PinnEngine(dev, L, H, 10000.0, alpha_b=10.0, alpha_e=1.0, precision=prec)
    E.net.set_flat(torch.tensor(flat))
    E.set_collocation(x, y)
    E.set_boundary(xb, yb, ub, vb)
    pred = torch.stack(E.predict(x, y)).cpu().numpy()          # [3][N]
    P = fr.unflatten(flat.astype(np.float64), 2, 3, L, H)
    out, _ = fr.forward1(P, x.astype(np.float64), y.astype(np.float64))
    tol = 2e-5 if prec == "bf16x3" else 3e-2
    assert np.abs(pred.T - out).max() < tol * max(1.0, np.abs(out).max())
    E.loss_and_grad()
    torch.cuda.synchronize()
    r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), 10000.0)
    b = fr.bc_loss_and_grad(P, xb.astype(np.float64), yb.astype(np.float64), ub, vb, alpha_b=10.0)
    g = E.grads.cpu().numpy()
    assert np.isfinite(g).all()
    assert _rel_l2(g, r["grad"] + b["grad"]) < (1e-4 if prec == "bf16x3" else 5e-2)


@pytest.mark.parametrize("H", [128, 256])
@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_64_column_tile_kernels_at_narrow_widths(H, prec, monkeypatch):
    """The 16-point-tile kernels (default only for fp32 at hidden = 256 and for wide nets) forced on
    for both precisions at hidden 128 / 256 through PINN_TILE_COLS."""
    monkeypatch.setenv("PINN_TILE_COLS", "64")
    eng = _engine_mod()
    L, N, Nb, Re = 3, 200, 50, 400.0
    flat = _rand_params(3, L, H, seed=3 + H)
    rng = np.random.RandomState(H)
    x = rng.rand(N).astype(np.float32); y = rng.rand(N).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[::40][:Nb].astype(np.float32) for a in ar.cavity_boundary())
    E = eng.PinnEngine(torch.device("cuda:0"), L, H, Re, alpha_b=10.0, alpha_e=1.0, precision=prec)
    E.net.set_flat(torch.tensor(flat)); E.set_collocation(x, y); E.set_boundary(xb, yb, ub, vb)
    assert E.plan_f.npad == ((N + 15) // 16) * 16          # 16-point tiles are in use
    E.loss_and_grad()
    P = fr.unflatten(flat.astype(np.float64), 2, 3, L, H)
    r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), Re)
    b = fr.bc_loss_and_grad(P, xb.astype(np.float64), yb.astype(np.float64), ub, vb, alpha_b=10.0)
    tol = 2e-5 if prec == "fp32" else 5e-4
    for k, name in enumerate(("eq1", "eq2", "eq3")):
        assert _rel_max(E.plan_f.field(name).cpu().numpy(), r["eqs"][k]) < tol, name
    assert _rel_l2(E.grads.cpu().numpy(), r["grad"] + b["grad"]) < 1e-4


def test_two_threads_alternating_nets_of_different_depth():
    """The library keeps no cached launch state (include/nsfnet_pinn.h): two host threads driving nets of
    different depth (different dynamic-LDS sizes of the same kernel template) on their own streams, in
    alternation, give the results of the serial runs."""
    import threading
    eng = _engine_mod()
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(12)
    x = rng.rand(700).astype(np.float32); y = rng.rand(700).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[::16].astype(np.float32) for a in ar.cavity_boundary())

    def make(L, prec):
        E = eng.PinnEngine(dev, L, 64, 300.0, alpha_b=10.0, alpha_e=1.0, precision=prec)
        E.net.set_flat(torch.tensor(_rand_params(3, L, 64, seed=L)))
        E.set_collocation(x, y); E.set_boundary(xb, yb, ub, vb)
        return E

    def run(E, out, k):
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            for _ in range(6):
                E.loss_and_grad()
            torch.cuda.current_stream().synchronize()
            out[k] = E.grads.cpu().numpy().copy()

    serial, threaded = {}, {}
    for prec in ("fp32", "bf16x3"):
        for L in (2, 9):            # created in alternating order: shallow, deep, shallow, deep
            run(make(L, prec), serial, (prec, L))
    engines = {(prec, L): make(L, prec) for prec in ("fp32", "bf16x3") for L in (9, 2)}
    ths = [threading.Thread(target=run, args=(E, threaded, k)) for k, E in engines.items()]
    for t in ths: t.start()
    for t in ths: t.join()
    for k in serial:
        assert np.array_equal(serial[k], threaded[k]), k


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16"])
def test_nan_weight_poisons_the_loss(prec):
    """A NaN parameter must surface as a NaN loss in every precision mode (the bf16 fragment copy uses the
    hardware conversion, which keeps NaNs; the integer rounding idiom turns some of them into 0 or inf)."""
    eng = _engine_mod()
    dev = torch.device("cuda:0")
    L, H = 3, 64
    xb, yb, ub, vb = (a.reshape(-1)[::16].astype(np.float32) for a in ar.cavity_boundary())
    x, y = (a.reshape(-1).astype(np.float32) for a in ar.uniform_grid(16, 16))
    for bits in (0x7FC00000, 0xFFFFFFFF, 0x7F800001):      # quiet NaN, all-ones NaN (-> +0 in the idiom), signalling NaN (-> inf)
        flat = _rand_params(3, L, H, seed=2)
        i = 3 * H + 5 * H + 7                                # an entry of layer_1.weight (goes through the bf16 copy)
        flat.view(np.uint32)[i] = bits
        E = eng.PinnEngine(dev, L, H, 100.0, alpha_b=10.0, alpha_e=1.0, precision=prec)
        E.net.set_flat(torch.tensor(flat))
        E.set_collocation(x, y); E.set_boundary(xb, yb, ub, vb)
        E.loss_and_grad()
        lt = E.loss_terms()
        assert np.isnan(float(lt["loss"])), (prec, hex(bits), float(lt["loss"]))
        assert np.isnan(float(lt["loss_b"])) and np.isnan(float(lt["loss_e"]))
