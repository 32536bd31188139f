"""GPU parity of the drop-in solver classes against vectors produced by the REFERENCE
itself (tests/golden/*.npz, see oracle/gen_golden.py): per-step residuals, loss terms,
gradients, the lagged artificial-viscosity sequence and parameters after Adam steps.

Tolerances (fp32; north star: per-step loss within 1e-4 relative of the reference):
  loss terms  rel <= 2e-5     residuals  max-abs <= 1e-4 * max|ref|     grads  rel-L2 <= 1e-4
  parameters after k Adam steps: update (p_k - p_0) rel-L2 <= 2e-3 (Adam's m/(sqrt(v)+eps)
  amplifies fp32 gradient noise on near-zero gradient entries).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _rel_max(a, b):
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _rel_l2(a, b):
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name", ["nsfnet_4x50_re100", "nsfnet_2x16_re1000"])
def test_nsfnet_solver_vs_reference_steps(golden_dir, name, tmp_path, monkeypatch):
    from nsfnet_amd import pinn_solver as ps
    from oracle import autograd_ref as ar
    monkeypatch.chdir(tmp_path)
    g = _load(golden_dir, name)
    P = ps.PysicsInformedNeuralNetwork(Re=float(g["Re"]), layers=int(g["L"]), hidden_size=int(g["H"]),
                                       N_f=int(g["N"]), bc_weight=float(g["alpha_b"]), eq_weight=float(g["alpha_e"]),
                                       learning_rate=float(g["lr"]), num_ins=2, num_outs=3)
    P.net.dev_net.set_flat(torch.tensor(g["w0"]))
    P.set_boundary_data(X=ar.cavity_boundary())
    P.set_eq_training_data(X=(g["x"], g["y"]))
    P.save_every = 0
    w0 = g["w0"].astype(np.float64)
    for k in range(g["losses"].shape[0]):
        loss, (loss_e, loss_b) = P.fwd_computing_loss_2d()
        mine = [loss.item(), loss_b.item(), P.loss_eq1.item(), P.loss_eq2.item(), P.loss_eq3.item()]
        np.testing.assert_allclose(mine, g["losses"][k], rtol=2e-5)
        if k == 0:
            assert _rel_max(_np(P.eq1_pred), g["eq1"]) < 1e-4
            assert _rel_max(_np(P.eq2_pred), g["eq2"]) < 1e-4
            assert _rel_max(_np(P.eq3_pred), g["eq3"]) < 1e-4
            assert _rel_l2(_np(P.engine.grads), g["grad0"]) < 1e-4
            np.testing.assert_allclose(_np(P.u_pred_b).reshape(-1), g["u_pred_b"].reshape(-1), atol=3e-6)
        P.engine.adam_step(P.opt.param_groups[0]["lr"])
        upd = _np(P.engine.net.params).astype(np.float64) - w0
        assert _rel_l2(upd, g["params_after"][k].astype(np.float64) - w0) < 2e-3


def test_nsfnet_solver_train_loop_and_checkpoint(golden_dir, tmp_path, monkeypatch, capsys):
    """train() drives the same steps (incl. log + checkpoint at step 0) and the checkpoint is a
    reference-format state_dict that the reference-style loader path accepts."""
    from nsfnet_amd import pinn_solver as ps
    from oracle import autograd_ref as ar
    monkeypatch.chdir(tmp_path)
    g = _load(golden_dir, "nsfnet_2x16_re1000")
    P = ps.PysicsInformedNeuralNetwork(Re=float(g["Re"]), layers=2, hidden_size=16, N_f=300, bc_weight=10.0, eq_weight=1.0)
    P.net.dev_net.set_flat(torch.tensor(g["w0"]))
    P.set_boundary_data(X=ar.cavity_boundary())
    P.set_eq_training_data(X=(g["x"], g["y"]))
    P.set_stage(1)
    P.train(num_epoch=5, lr=float(g["lr"]))
    w0 = g["w0"].astype(np.float64)
    assert _rel_l2(_np(P.engine.net.params) - w0, g["params_after"][4] - w0) < 2e-3
    ck = tmp_path / "results" / "Re1000.0" / "2x16_Nf0k_lamB10.01" / "model_cavity_loop_0.pth"
    assert ck.exists()
    sd = torch.load(str(ck), weights_only=True)
    assert list(sd.keys()) == ["layers.layer_0.weight", "layers.layer_0.bias", "layers.layer_1.weight",
                               "layers.layer_1.bias", "layers.layer_2.weight", "layers.layer_2.bias"]
    assert tuple(sd["layers.layer_1.weight"].shape) == (16, 16)
    # checkpoint at step 0 is written AFTER the first optimizer step (NSFnet/pinn_solver.py:251-276)
    flat = torch.cat([v.reshape(-1) for v in sd.values()]).numpy()
    assert _rel_l2(flat - w0, g["params_after"][0] - w0) < 2e-3
    P2 = ps.PysicsInformedNeuralNetwork(Re=1000.0, layers=2, hidden_size=16, net_params=str(ck))
    np.testing.assert_array_equal(_np(P2.engine.net.params), flat)
    assert "eq1_loss" in capsys.readouterr().out


def test_nsfnet_6x256_vs_reference(golden_dir, tmp_path, monkeypatch):
    from nsfnet_amd import pinn_solver as ps
    from oracle import autograd_ref as ar
    monkeypatch.chdir(tmp_path)
    g = _load(golden_dir, "nsfnet_6x256_re2000_n256")
    torch.manual_seed(int(g["seed"]))          # same RNG stream as the reference constructor
    P = ps.PysicsInformedNeuralNetwork(Re=2000, layers=6, hidden_size=256, N_f=256, bc_weight=10, eq_weight=1)
    s = int(g["stride"])
    np.testing.assert_array_equal(_np(P.engine.net.params)[::s], g["w0_sample"])
    w0 = _np(P.engine.net.params).astype(np.float64)
    P.set_boundary_data(X=ar.cavity_boundary())
    P.set_eq_training_data(X=(g["x"], g["y"]))
    loss, (loss_e, loss_b) = P.fwd_computing_loss_2d()
    mine = [loss.item(), loss_b.item(), P.loss_eq1.item(), P.loss_eq2.item(), P.loss_eq3.item()]
    np.testing.assert_allclose(mine, g["losses"][0], rtol=2e-5)
    assert _rel_max(_np(P.eq1_pred), g["eq1"]) < 1e-4
    assert _rel_l2(_np(P.engine.grads)[::s], g["grad0"]) < 1e-4
    P.engine.adam_step(1e-3)
    assert _rel_l2(_np(P.engine.net.params)[::s] - w0[::s], g["params_after"][0] - w0[::s]) < 2e-3


def _make_ev(g, coord_scale=1.0):
    from nsfnet_amd import ev_pinn_solver as es
    P = es.PysicsInformedNeuralNetwork(Re=float(g["Re"]), layers=int(g["L"]), layers_1=int(g["L1"]),
                                       hidden_size=int(g["H"]), hidden_size_1=int(g["H1"]), N_f=int(g["N"]),
                                       alpha_evm=float(g["alpha_evm"]), bc_weight=float(g["alpha_b"]),
                                       eq_weight=float(g["alpha_e"]), learning_rate=float(g["lr"]),
                                       supervised_data_weight=0.0)
    return P


@pytest.mark.parametrize("name", ["ev_4x50_4x40_re4000", "ev_2x16_sdf_scaled"])
def test_ev_solver_vs_reference_steps(golden_dir, name, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    g = _load(golden_dir, name)
    P = _make_ev(g)
    P.net.dev_net.set_flat(torch.tensor(g["w0"])); P.net_1.dev_net.set_flat(torch.tensor(g["w0_e"]))
    # same call order as ev-NSFnet/train.py:139-144
    P.set_boundary_data(X=(g["x_b"], g["y_b"], g["u_b"], g["v_b"]))
    P.set_coordinate_transform(float(g["coord_scale"]))
    w = g["weights"] if "weights" in g.files else None
    P.set_eq_training_data(X=(g["x"], g["y"]), weights=w)
    np.testing.assert_allclose(_np(P.vis_t_minus).reshape(-1), g["vis_t_minus0"].reshape(-1), rtol=2e-5, atol=1e-9)
    P.freeze_evm_net(0)                      # solve_Adam prologue (:452)
    w0 = g["w0"].astype(np.float64)
    for k in range(g["losses"].shape[0]):
        P._apply_freeze_schedule(k)          # re-creates Adam at epoch 1 exactly as the reference loop does
        loss, (loss_e, loss_b) = P.fwd_computing_loss_2d()
        mine = [loss.item(), loss_b.item(), P.loss_eq1.item(), P.loss_eq2.item(), P.loss_eq3.item(), P.loss_eq4.item()]
        np.testing.assert_allclose(mine, g["losses"][k], rtol=3e-5)
        np.testing.assert_allclose(_np(P.vis_t).reshape(-1), g["vis_t"][k], rtol=3e-5, atol=1e-9)
        if k == 0:
            for key, t in (("eq1", P.eq1_pred), ("eq2", P.eq2_pred), ("eq3", P.eq3_pred), ("eq4", P.eq4_pred)):
                assert _rel_max(_np(t), g[key]) < 1e-4, key
            assert _rel_l2(_np(P.engine.grads), g["grad0"]) < 1e-4
        P.engine.adam_step(P.opt.param_groups[0]["lr"])
        upd = _np(P.engine.net.params).astype(np.float64) - w0
        assert _rel_l2(upd, g["params_after"][k].astype(np.float64) - w0) < 2e-3
    np.testing.assert_array_equal(_np(P.engine.net_e.params), g["params_e_after"])     # frozen: untouched


def test_ev_freeze_schedule_vs_reference(golden_dir, tmp_path, monkeypatch):
    """Steps 10000..10002 of the reference loop: the entropy net moves for exactly one step and
    Adam is re-created twice (ev-NSFnet/pinn_solver.py:459-462, 489-511)."""
    monkeypatch.chdir(tmp_path)
    g = _load(golden_dir, "ev_freeze_2x8")
    P = _make_ev(g)
    P.net.dev_net.set_flat(torch.tensor(g["p_10000"])); P.net_1.dev_net.set_flat(torch.tensor(g["pe_10000"]))
    P.set_boundary_data(X=(g["x_b"], g["y_b"], g["u_b"], g["v_b"]))
    P.set_eq_training_data(X=(g["x"], g["y"]))
    P.engine.plan_f.vis_t_minus = torch.tensor(g["vtm_10000"].reshape(-1)).to(P.device)
    P.freeze_evm_net(0)
    for k in (10000, 10001, 10002):
        P._apply_freeze_schedule(k)
        loss, _ = P.fwd_computing_loss_2d()
        assert abs(loss.item() - float(g["loss_%d" % k])) <= 1e-4 * abs(float(g["loss_%d" % k]))
        P.engine.adam_step(P.opt.param_groups[0]["lr"])
        p_ref, pe_ref = g["p_%d" % (k + 1)], g["pe_%d" % (k + 1)]
        p_prev, pe_prev = g["p_%d" % k], g["pe_%d" % k]
        assert _rel_l2(_np(P.engine.net.params) - p_prev, p_ref - p_prev) < 5e-3
        if k == 10000:
            assert _rel_l2(_np(P.engine.net_e.params) - pe_prev, pe_ref - pe_prev) < 5e-3
            assert np.abs(_np(P.engine.net_e.params) - pe_prev).max() > 1e-4
        else:
            np.testing.assert_allclose(_np(P.engine.net_e.params), pe_ref, atol=2e-6)


def test_ev_train_loop_runs_and_logs(golden_dir, tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    g = _load(golden_dir, "ev_2x16_sdf_scaled")
    P = _make_ev(g)
    P.net.dev_net.set_flat(torch.tensor(g["w0"])); P.net_1.dev_net.set_flat(torch.tensor(g["w0_e"]))
    P.log_interval = 2
    P.current_stage = "Stage 1"
    P.set_boundary_data(X=(g["x_b"], g["y_b"], g["u_b"], g["v_b"]))
    P.set_coordinate_transform(float(g["coord_scale"]))
    P.set_eq_training_data(X=(g["x"], g["y"]), weights=g["weights"])
    P.set_alpha_evm(float(g["alpha_evm"]))
    P.train(num_epoch=4, lr=float(g["lr"]))
    w0 = g["w0"].astype(np.float64)
    assert _rel_l2(_np(P.engine.net.params) - w0, g["params_after"][3] - w0) < 2e-3
    out = capsys.readouterr().out
    assert "throughput=" in out and "Re_eff=" in out
    d = tmp_path / "results" / ("Re%s" % float(g["Re"]))
    assert any(f.name.endswith("_evm") for f in d.rglob("*"))
    eu, ev_, ep = P.evaluate(g["x"], g["y"], g["x"] * 0 + 1.0, g["y"] * 0 + 1.0, g["x"] * 0 + 1.0)
    assert np.isfinite([eu, ev_, ep]).all()


@pytest.mark.parametrize("name", ["nsfnet_4x50_re100", "nsfnet_6x256_re2000_n256"])
def test_bf16x3_solver_loss_within_1e4_of_reference(golden_dir, name, tmp_path, monkeypatch):
    """North-star bar for the bf16 MFMA path: per-step loss within 1e-4 relative of the reference."""
    from nsfnet_amd import pinn_solver as ps
    from oracle import autograd_ref as ar
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("NSFNET_PRECISION", "bf16x3")
    g = _load(golden_dir, name)
    torch.manual_seed(int(g["seed"]))
    P = ps.PysicsInformedNeuralNetwork(Re=float(g["Re"]), layers=int(g["L"]), hidden_size=int(g["H"]), N_f=int(g["N"]),
                                       bc_weight=float(g["alpha_b"]), eq_weight=float(g["alpha_e"]))
    assert P.engine.net.precision == ("bf16x3",) * 3
    if "w0" in g.files:
        P.net.dev_net.set_flat(torch.tensor(g["w0"]))
    P.set_boundary_data(X=ar.cavity_boundary())
    P.set_eq_training_data(X=(g["x"], g["y"]))
    P.save_every = 0
    for k in range(g["losses"].shape[0]):
        loss, (loss_e, loss_b) = P.fwd_computing_loss_2d()
        mine = [loss.item(), loss_b.item(), P.loss_eq1.item(), P.loss_eq2.item(), P.loss_eq3.item()]
        np.testing.assert_allclose(mine, g["losses"][k], rtol=1e-4)
        P.engine.adam_step(P.opt.param_groups[0]["lr"])


def test_nsfnet_solver_l2_loss_mode_vs_reference(golden_dir, tmp_path, monkeypatch):
    """fwd_computing_loss_2d(loss_mode='L2') of the drop-in class (NSFnet/pinn_solver.py:202-204, 214-217) against the
    reference's own branch: loss, its two parts, the gradient and three Adam steps.  Same HIP kernels as the MSE mode,
    adjoint coefficients alpha / ||r_k|| from one host read of the forward sums."""
    from nsfnet_amd import pinn_solver as ps
    monkeypatch.chdir(tmp_path)
    g = _load(golden_dir, "nsfnet_l2_3x24_re400")
    P = ps.PysicsInformedNeuralNetwork(Re=float(g["Re"]), layers=int(g["L"]), hidden_size=int(g["H"]),
                                       N_f=int(g["N"]), bc_weight=float(g["alpha_b"]), eq_weight=float(g["alpha_e"]),
                                       learning_rate=float(g["lr"]), num_ins=2, num_outs=3)
    P.net.dev_net.set_flat(torch.tensor(g["w0"]))
    P.set_boundary_data(X=(g["x_b"], g["y_b"], g["u_b"], g["v_b"]))
    P.set_eq_training_data(X=(g["x"], g["y"]))
    P.save_every = 0
    w0 = g["w0"].astype(np.float64)
    for k in range(g["losses"].shape[0]):
        loss, (loss_e, loss_b) = P.fwd_computing_loss_2d(loss_mode='L2')
        np.testing.assert_allclose([loss.item(), loss_b.item(), loss_e.item()], g["losses"][k], rtol=2e-5)
        if k == 0:
            assert _rel_l2(_np(P.engine.grads), g["grad0"]) < 1e-4
        P.engine.adam_step(P.opt.param_groups[0]["lr"])
        upd = _np(P.engine.net.params).astype(np.float64) - w0
        assert _rel_l2(upd, g["params_after"][k].astype(np.float64) - w0) < 2e-3
    with pytest.raises(ValueError):
        P.fwd_computing_loss_2d(loss_mode='L1')
