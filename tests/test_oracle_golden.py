"""Pin the CPU oracle (oracle/) against vectors produced by the reference itself
(tests/golden/*.npz, written by oracle/gen_golden.py from the upstream code).

The reference ships no tests or golden vectors for this path,
so these reference-generated fixtures are what pins parity.
"""
import os

import numpy as np
import pytest
import torch

from oracle import autograd_ref as ar
from oracle import fwdmode_ref as fr


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _net_from_flat(flat, n_out, L, H, dtype=torch.float32):
    net = ar.RefFCNet(2, n_out, L, H).to(dtype)
    off = 0
    with torch.no_grad():
        for p in net.parameters():
            n = p.numel()
            p.copy_(torch.tensor(flat[off:off + n]).reshape(p.shape).to(dtype))
            off += n
    assert off == len(flat)
    return net


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_boundary_set_matches_reference(golden_dir):
    g = _load(golden_dir, "nsfnet_4x50_re100")
    xb, yb, ub, vb = ar.cavity_boundary()
    for mine, ref in zip((xb, yb, ub, vb), (g["x_b"], g["y_b"], g["u_b"], g["v_b"])):
        assert mine.shape == ref.shape == (2052, 1)
        np.testing.assert_allclose(mine, ref, rtol=0, atol=1e-15)
    assert abs(ub.max() - 0.986525) < 1e-6


@pytest.mark.parametrize("name", ["nsfnet_4x50_re100", "nsfnet_2x16_re1000"])
def test_autograd_restatement_vs_reference_steps(golden_dir, name):
    g = _load(golden_dir, name)
    L, H, Re = int(g["L"]), int(g["H"]), float(g["Re"])
    net = _net_from_flat(g["w0"], 3, L, H)
    o = ar.NSFnetOracle(net, Re, alpha_b=float(g["alpha_b"]), alpha_e=float(g["alpha_e"]), lr=float(g["lr"]))
    o.set_data(g["x"], g["y"], *ar.cavity_boundary())
    for k in range(g["losses"].shape[0]):
        total = o.step()
        ref = g["losses"][k]
        mine = [total, float(o.loss_b.detach())] + [float(v.detach()) for v in o.loss_eq]
        np.testing.assert_allclose(mine, ref, rtol=2e-6)
        if k == 0:
            assert _rel(o.eq1.detach().numpy(), g["eq1"]) < 1e-5
            assert _rel(o.eq2.detach().numpy(), g["eq2"]) < 1e-5
            assert _rel(o.eq3.detach().numpy(), g["eq3"]) < 1e-5
            assert _rel(o.grads.numpy(), g["grad0"]) < 1e-5
        assert _rel(ar.flat_params(net).numpy(), g["params_after"][k]) < 1e-5


def test_autograd_restatement_6x256(golden_dir):
    g = _load(golden_dir, "nsfnet_6x256_re2000_n256")
    net = ar.seeded_net(3, 6, 256, seed=int(g["seed"]))
    w0 = ar.flat_params(net).numpy()
    s = int(g["stride"])
    np.testing.assert_array_equal(w0[::s], g["w0_sample"])      # same torch init stream as the reference
    o = ar.NSFnetOracle(net, float(g["Re"]), alpha_b=float(g["alpha_b"]), alpha_e=float(g["alpha_e"]))
    o.set_data(g["x"], g["y"], *ar.cavity_boundary())
    total = o.step()
    np.testing.assert_allclose([total, float(o.loss_b)] + [float(v) for v in o.loss_eq], g["losses"][0], rtol=5e-6)
    assert _rel(o.grads.numpy()[::s], g["grad0"]) < 2e-5
    assert _rel(o.eq1.detach().numpy(), g["eq1"]) < 2e-5


@pytest.mark.parametrize("name", ["ev_4x50_4x40_re4000", "ev_2x16_sdf_scaled"])
def test_ev_restatement_vs_reference_steps(golden_dir, name):
    g = _load(golden_dir, name)
    net = _net_from_flat(g["w0"], 3, int(g["L"]), int(g["H"]))
    net_e = _net_from_flat(g["w0_e"], 1, int(g["L1"]), int(g["H1"]))
    o = ar.EvNSFnetOracle(net, net_e, float(g["Re"]), float(g["alpha_evm"]), alpha_b=float(g["alpha_b"]),
                          alpha_e=float(g["alpha_e"]), lr=float(g["lr"]), coord_scale=float(g["coord_scale"]))
    w = g["weights"] if "weights" in g.files else None
    o.set_data(g["x"], g["y"], g["x_b"], g["y_b"], g["u_b"], g["v_b"], weights=w)
    np.testing.assert_allclose(o.vis_t_minus.numpy(), g["vis_t_minus0"], rtol=1e-6, atol=1e-9)
    for k in range(g["losses"].shape[0]):
        total = o.step(epoch_id=k)
        mine = [total, float(o.loss_b.detach())] + [float(v.detach()) for v in o.loss_eq]
        np.testing.assert_allclose(mine, g["losses"][k], rtol=5e-6)
        np.testing.assert_allclose(o.vis_t.numpy().reshape(-1), g["vis_t"][k], rtol=1e-5, atol=1e-9)
        if k == 0:
            for i, key in enumerate(("eq1", "eq2", "eq3", "eq4")):
                assert _rel(o.eq[i].detach().numpy(), g[key]) < 1e-5
            assert _rel(o.grads.numpy(), g["grad0"]) < 1e-5
        assert _rel(ar.flat_params(net).numpy(), g["params_after"][k]) < 1e-5
    np.testing.assert_array_equal(ar.flat_params(net_e).numpy(), g["params_e_after"])   # frozen: untouched


@pytest.mark.parametrize("name", ["ev_sup_3x24_nanp", "ev_sup_3x24_nop", "ev_sup_3x24_allnanp"])
def test_ev_supervised_loss_vs_reference(golden_dir, name):
    """loss_s with NaN-masked pressure targets (ev-NSFnet/pinn_solver.py:399-411), its gradient and the Adam
    steps it drives, against the reference's own numbers."""
    g = _load(golden_dir, name)
    net = _net_from_flat(g["w0"], 3, int(g["L"]), int(g["H"]))
    net_e = _net_from_flat(g["w0_e"], 1, int(g["L1"]), int(g["H1"]))
    o = ar.EvNSFnetOracle(net, net_e, float(g["Re"]), float(g["alpha_evm"]), alpha_b=float(g["alpha_b"]),
                          alpha_e=float(g["alpha_e"]), lr=float(g["lr"]), alpha_s=float(g["alpha_s"]))
    o.set_data(g["x"], g["y"], g["x_b"], g["y_b"], g["u_b"], g["v_b"])
    p_s = g["p_s"] if "p_s" in g.files else None
    o.set_supervised(g["x_s"], g["y_s"], g["u_s"], g["v_s"], p_s)
    if name.endswith("_nanp"):
        assert 0 < np.isnan(p_s).sum() < p_s.size          # the fixture really exercises the mask
    vtm0 = o.vis_t_minus.clone()
    for k in range(g["losses"].shape[0]):
        total = o.step(epoch_id=k)
        mine = [total, float(o.loss_b.detach()), float(o.loss_e.detach()), float(o.loss_s.detach())]
        np.testing.assert_allclose(mine, g["losses"][k], rtol=5e-6)
        if k == 0:
            assert _rel(o.grads.numpy(), g["grad0"]) < 1e-5
        assert _rel(ar.flat_params(net).numpy(), g["params_after"][k]) < 1e-5
    # weight switched to 0: the branch is skipped and loss_s reads 0 (:253-255, :398-400)
    net = _net_from_flat(g["w0"], 3, int(g["L"]), int(g["H"]))
    o = ar.EvNSFnetOracle(net, net_e, float(g["Re"]), float(g["alpha_evm"]), alpha_b=float(g["alpha_b"]),
                          alpha_e=float(g["alpha_e"]), alpha_s=0.0)
    o.set_data(g["x"], g["y"], g["x_b"], g["y_b"], g["u_b"], g["v_b"])
    o.vis_t_minus = vtm0
    o.set_supervised(g["x_s"], g["y_s"], g["u_s"], g["v_s"], p_s)
    total = float(o.loss().detach())
    np.testing.assert_allclose([total, float(o.loss_s)], g["loss_alpha0"], rtol=5e-6)


def test_ev_freeze_schedule(golden_dir):
    """Steps 10000..10002: the entropy net trains for exactly one step and Adam
    restarts twice (ev-NSFnet/pinn_solver.py:459-462, 489-511)."""
    g = _load(golden_dir, "ev_freeze_2x8")
    net = _net_from_flat(g["p_10000"], 3, int(g["L"]), int(g["H"]))
    net_e = _net_from_flat(g["pe_10000"], 1, int(g["L1"]), int(g["H1"]))
    o = ar.EvNSFnetOracle(net, net_e, float(g["Re"]), float(g["alpha_evm"]), alpha_b=float(g["alpha_b"]),
                          alpha_e=float(g["alpha_e"]), lr=float(g["lr"]))
    o.set_data(g["x"], g["y"], g["x_b"], g["y_b"], g["u_b"], g["v_b"])
    o.vis_t_minus = torch.tensor(g["vtm_10000"])
    for k in (10000, 10001, 10002):
        total = o.step(epoch_id=k)
        assert abs(total - float(g["loss_%d" % k])) <= 2e-5 * abs(float(g["loss_%d" % k]))
        assert _rel(ar.flat_params(net).numpy(), g["p_%d" % (k + 1)]) < 2e-5
        assert _rel(ar.flat_params(net_e).numpy(), g["pe_%d" % (k + 1)]) < 2e-5
    assert np.abs(g["pe_10001"] - g["pe_10000"]).max() > 1e-4      # moved once
    np.testing.assert_array_equal(g["pe_10002"], g["pe_10001"])     # then frozen again


# ---- forward-mode restatement (the kernels' executable spec) ---------------
def test_fwdmode_matches_autograd_fp64():
    torch.manual_seed(5)
    net = ar.RefFCNet(2, 3, 3, 24).double()
    net_e = ar.RefFCNet(2, 1, 2, 10).double()
    rng = np.random.RandomState(0)
    x, y = rng.rand(200, 1), rng.rand(200, 1)
    xb, yb, ub, vb = (a[::16] for a in ar.cavity_boundary())
    w = 0.5 + rng.rand(200)
    o = ar.EvNSFnetOracle(net, net_e, 500.0, 0.05, alpha_b=10.0, alpha_e=1.0, coord_scale=2.0)
    o.defreeze_e()
    o.set_data(x, y, xb, yb, ub, vb, weights=w)
    vtm = o.vis_t_minus.numpy().reshape(-1).copy()
    o.step()
    # parameters BEFORE the step are gone after opt.step(); rebuild from a fresh copy
    torch.manual_seed(5)
    net2 = ar.RefFCNet(2, 3, 3, 24).double(); net_e2 = ar.RefFCNet(2, 1, 2, 10).double()
    P = fr.unflatten(ar.flat_params(net2).numpy(), 2, 3, 3, 24)
    Pe = fr.unflatten(ar.flat_params(net_e2).numpy(), 2, 1, 2, 10)
    e, saved_e = fr.forward1(Pe, x, y)
    vis_t = np.minimum(20.0 / 500.0, vtm)
    r = fr.pde_loss_and_grad(P, x, y, 500.0, alpha_e=1.0, vis_t=vis_t, e=e[:, 0], w=w, scale=2.0)
    b = fr.bc_loss_and_grad(P, xb, yb, ub, vb, alpha_b=10.0)
    for k in range(4):
        assert _rel(r["eqs"][k], o.eq[k].detach().numpy().reshape(-1)) < 1e-12
        assert abs(r["sums"][k] / 200 - float(o.loss_eq[k])) < 1e-12 * max(1.0, float(o.loss_eq[k]))
    assert abs(sum(b["sums"]) / len(xb) - float(o.loss_b)) < 1e-13
    assert _rel(r["grad"] + b["grad"], o.grads.numpy()) < 1e-11
    ge = fr.backward1(Pe, x, y, saved_e, r["e_adj"].reshape(-1, 1))
    assert _rel(ge, o.grads_e.numpy()) < 1e-11


def test_fwdmode_vs_reference_fixture_fp32_inputs(golden_dir):
    g = _load(golden_dir, "nsfnet_4x50_re100")
    P = fr.unflatten(g["w0"].astype(np.float64), 2, 3, 4, 50)
    x32, y32 = g["x"].astype(np.float32).astype(np.float64), g["y"].astype(np.float32).astype(np.float64)
    xb, yb, ub, vb = (a.astype(np.float32).astype(np.float64) for a in (g["x_b"], g["y_b"], g["u_b"], g["v_b"]))
    r = fr.pde_loss_and_grad(P, x32, y32, 100.0, alpha_e=1.0)
    b = fr.bc_loss_and_grad(P, xb, yb, ub, vb, alpha_b=10.0)
    # the fixture is fp32 arithmetic; fp64 truth must sit within fp32 noise of it
    assert _rel(r["eqs"][0], g["eq1"].reshape(-1)) < 2e-5
    assert _rel(r["eqs"][2], g["eq3"].reshape(-1)) < 2e-5
    N = len(x32)
    tot = 10.0 * sum(b["sums"]) / 2052 + sum(r["sums"]) / N
    assert abs(tot - g["losses"][0, 0]) < 2e-6 * tot
    assert _rel(r["grad"] + b["grad"], g["grad0"]) < 5e-5


def test_adam_restatement(golden_dir):
    g = _load(golden_dir, "nsfnet_2x16_re1000")
    # grad0 + w0 -> params_after[0] with fresh Adam state at step 1
    p, m, v = fr.adam_step(g["w0"].astype(np.float64), g["grad0"].astype(np.float64), 0.0, 0.0, 1, float(g["lr"]))
    assert _rel(p, g["params_after"][0]) < 1e-6



def test_autograd_restatement_l2_loss_mode_vs_reference(golden_dir):
    """loss_mode='L2' (NSFnet/pinn_solver.py:202-204, 214-217: 2-norms of the boundary misfit and the residuals instead of
    mean squares) - fixture generated by the reference's own branch (oracle/gen_golden.py gen_nsfnet_l2)."""
    g = _load(golden_dir, "nsfnet_l2_3x24_re400")
    L, H, Re = int(g["L"]), int(g["H"]), float(g["Re"])
    net = _net_from_flat(g["w0"], 3, L, H)
    o = ar.NSFnetOracle(net, Re, alpha_b=float(g["alpha_b"]), alpha_e=float(g["alpha_e"]), lr=float(g["lr"]), loss_mode="L2")
    o.set_data(g["x"], g["y"], g["x_b"], g["y_b"], g["u_b"], g["v_b"])
    for k in range(g["losses"].shape[0]):
        total = o.step()
        np.testing.assert_allclose([total, float(o.loss_b.detach()), float(o.loss_e.detach())], g["losses"][k], rtol=2e-6)
        if k == 0:
            assert _rel(o.grads.numpy(), g["grad0"]) < 1e-5
        assert _rel(ar.flat_params(net).numpy(), g["params_after"][k]) < 1e-5
