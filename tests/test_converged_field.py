"""Converged-field gate: the ev-NSFnet weights produced by THIS engine with the reference's
production schedule shape (scripts/converge_ev.py, 2.7 M Adam steps on one MI355X)
reproduce the DNS cavity flow at Re = 3000 and Re = 2000 to the "< 4 %" relative L2 error the reference's README
quotes.  The checkpoint is in the reference's state_dict format and the DNS field is the file the
reference ships; evaluation goes through the drop-in solver's evaluate() (HIP value-mode forward), in
every precision mode."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("Re,dns,final", [(3000, "cavity_Re3000_256_Uniform.mat", 3.44), (2000, "cavity_Re2000_256.mat", 2.09)])
@pytest.mark.parametrize("prec,bar", [("fp32", 4.0), ("bf16x3", 4.0), ("bf16", 6.0)])
def test_trained_ev_nsfnet_matches_dns(prec, bar, Re, dns, final, monkeypatch, tmp_path, capsys):
    monkeypatch.setenv("NSFNET_PRECISION", prec)
    monkeypatch.chdir(tmp_path)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    from nsfnet_amd import ev_pinn_solver as es, cavity_data as cavity
    P = es.PysicsInformedNeuralNetwork(
        Re=Re, layers=6, layers_1=4, hidden_size=80, hidden_size_1=40, N_f=1000, alpha_evm=0.002,
        net_params=os.path.join(HERE, "golden", "trained", "ev_re%d_6x80_net.pth" % Re),
        net_params_1=os.path.join(HERE, "golden", "trained", "ev_re%d_4x40_evm.pth" % Re))
    star = cavity.EvDataLoader(N_f=1000).loading_evaluate_data(
        os.path.join(HERE, "golden", "dns", dns))
    assert star[0].shape[0] == 257 * 257
    eu, ev, ep = P.evaluate(*star)
    assert "Error u" in capsys.readouterr().out
    assert eu < bar and ev < bar, (eu, ev)
    if prec != "bf16":
        assert abs(eu - final) < 0.05 and abs(ev - final) < 0.05    # the run's own end-of-training report


def test_config3_shape_run_matches_dns(monkeypatch, tmp_path):
    """The headline shape itself (BASELINE config 3's 6x256 net on 360 000 collocation points, ev flavour with the 4x40
    entropy net, Re = 2000) trained by THIS engine in bf16x3 on the role-split kernels: six stages of the production
    schedule at 0.21x plus five repeats of the last one (1 155 000 steps, 167 GPU-minutes on one MI355X,
    profiles/r02_convergence_ev_config3shape_re2000.jsonl: 55.8 -> 24.7 -> 16.1 -> 10.4 -> 7.8 -> 6.2 -> 5.2 -> 4.5 -> 4.0 -> 3.6 -> 3.3 %;
    the last 525 000 steps on the 24-bit-spill builds, 8.1-8.6 ms/step) - inside the "< 4 %" the reference's README quotes, at
    0.39 of the production schedule's steps.  The bars here are the run's own end-of-training report and the flow topology
    (one primary vortex where the DNS has it)."""
    monkeypatch.setenv("NSFNET_PRECISION", "bf16x3")
    monkeypatch.chdir(tmp_path)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "scripts"))
    import flow_topology as ft
    from nsfnet_amd import ev_pinn_solver as es, cavity_data as cavity
    P = es.PysicsInformedNeuralNetwork(
        Re=2000, layers=6, layers_1=4, hidden_size=256, hidden_size_1=40, N_f=1000, alpha_evm=0.002,
        net_params=os.path.join(HERE, "golden", "trained", "ev_re2000_6x256_net.pth"),
        net_params_1=os.path.join(HERE, "golden", "trained", "ev_re2000_6x256_evm.pth"))
    star = cavity.EvDataLoader(N_f=1000).loading_evaluate_data(os.path.join(HERE, "golden", "dns", "cavity_Re2000_256.mat"))
    eu, ev, ep = P.evaluate(*star)
    assert abs(eu - 3.29) < 0.1 and abs(ev - 3.33) < 0.1, (eu, ev)
    assert eu < 4.0 and ev < 4.0
    X, Y, U, V = ft.load_dns(os.path.join(HERE, "golden", "dns", "cavity_Re2000_256.mat"))
    u, v = ft.predict_field("ev", os.path.join(HERE, "golden", "trained", "ev_re2000_6x256_net.pth"), X, Y, 6, 256, Re=2000.0)
    t_net, t_dns = ft.topology(X, Y, u, v), ft.topology(X, Y, U, V)
    assert abs(t_net["primary"]["x"] - t_dns["primary"]["x"]) < 0.03 and abs(t_net["primary"]["y"] - t_dns["primary"]["y"]) < 0.03, (
        t_net["primary"], t_dns["primary"])


def test_config4_dns_file_evaluate_and_test_on_the_385_grid(monkeypatch, tmp_path, capsys):
    """BASELINE config 4 names cavity_Re4000_384_Uniform.mat: a 385 x 385 grid, the case that breaks the reference's
    hard-coded 257 x 257 reshape in test() (ev-NSFnet/pinn_solver.py:723-726; NSFnet/pinn_solver.py:348-350).  evaluate()
    and test() of the ev drop-in class on that file, config 4's net shape (6x256 + 4x40): the grid shape comes from the
    data and the saved .mat holds 385 x 385 fields.  (Seeded weights - no Re = 4000 run of this engine exists yet, so the
    errors themselves are not a statement; the trained-weights gates are the tests above.)"""
    import scipy.io
    monkeypatch.setenv("NSFNET_PRECISION", "bf16x3")
    monkeypatch.chdir(tmp_path)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    torch.manual_seed(4000)
    from nsfnet_amd import ev_pinn_solver as es, cavity_data as cavity
    P = es.PysicsInformedNeuralNetwork(Re=4000, layers=6, layers_1=4, hidden_size=256, hidden_size_1=40, N_f=1000,
                                       alpha_evm=0.03)
    star = cavity.EvDataLoader(N_f=1000).loading_evaluate_data(os.path.join(HERE, "golden", "dns", "cavity_Re4000_384_Uniform.mat"))
    assert star[0].shape[0] == 385 * 385
    eu, ev, ep = P.evaluate(*star)
    assert np.isfinite([eu, ev, ep]).all() and "Error u" in capsys.readouterr().out
    out_dir = str(tmp_path / "res")
    tu, tv, tp = P.test(*star, loop=0, save_dir=out_dir)
    assert abs(tu - eu) < 1e-6 and abs(tv - ev) < 1e-6
    m = scipy.io.loadmat(os.path.join(out_dir, "cavity_result_loop_0.mat"))
    for key in ("U_pred", "V_pred", "P_pred", "E_pred"):
        assert m[key].shape == (385, 385), (key, m[key].shape)
    # the saved field is the network's prediction on the file's own grid, row for row
    u_pred = P.predict(None, (star[0], star[1]))[0]
    u_pred = u_pred.detach().cpu().numpy() if hasattr(u_pred, "detach") else np.asarray(u_pred)
    np.testing.assert_allclose(m["U_pred"].reshape(-1), u_pred.reshape(-1), rtol=0, atol=1e-6)


def test_config4_shape_run_vs_dns(monkeypatch, tmp_path):
    """BASELINE config 4's shape trained by THIS engine on its own data (profiles/r03_convergence_ev_config4shape_re4000.jsonl):
    ev-NSFnet, Re = 4000, 6x256 + 4x40 nets, 250 000 LHS points (config 4's per-GPU share), SDF weights, bf16x3 on the
    role-split kernels, six stages of the production schedule at 0.30x plus eight repeats of the last one = 2.1 M steps, 201
    GPU-minutes on one MI355X.  Relative L2 error of (u, v) against the reference's cavity_Re4000_384_Uniform.mat after each
    slice: 63.0 -> 39.4 -> 35.0 -> 30.9 -> 27.8 -> 25.2 -> 23.1 -> 21.3 -> 19.9 -> 18.6 -> 17.5 -> 16.5 -> 15.6 -> 14.8 %, still
    falling 0.8 points per 150 000 steps at 0.7 of the schedule's step count - NOT yet the "< 4 %" of the reference's README, and the test says what it
    is: the run's own end-of-training report reproduced from the kept weights, on the 385 x 385 grid."""
    monkeypatch.setenv("NSFNET_PRECISION", "bf16x3")
    monkeypatch.chdir(tmp_path)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    from nsfnet_amd import ev_pinn_solver as es, cavity_data as cavity
    P = es.PysicsInformedNeuralNetwork(
        Re=4000, layers=6, layers_1=4, hidden_size=256, hidden_size_1=40, N_f=1000, alpha_evm=0.002,
        net_params=os.path.join(HERE, "golden", "trained", "ev_re4000_6x256_net.pth"),
        net_params_1=os.path.join(HERE, "golden", "trained", "ev_re4000_6x256_evm.pth"))
    star = cavity.EvDataLoader(N_f=1000).loading_evaluate_data(os.path.join(HERE, "golden", "dns", "cavity_Re4000_384_Uniform.mat"))
    assert star[0].shape[0] == 385 * 385
    eu, ev, ep = P.evaluate(*star)
    assert abs(eu - 14.80) < 0.2 and abs(ev - 14.77) < 0.2, (eu, ev)
