"""The hipGraph-replayed training step must be the SAME computation as the eager launch
sequence: identical kernels in identical order, so parameters agree bit for bit - also
across the ev schedule's Adam resets / freeze switches and across an lr change."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_nsfnet(monkeypatch, graph, steps):
    from nsfnet_amd import pinn_solver as ps
    from oracle import autograd_ref as ar
    monkeypatch.setenv("NSFNET_GRAPH", "1" if graph else "0")
    torch.manual_seed(3)
    P = ps.PysicsInformedNeuralNetwork(Re=400.0, layers=3, hidden_size=40, N_f=900, bc_weight=10.0,
                                       eq_weight=1.0, learning_rate=1e-3, num_ins=2, num_outs=3)
    x, y = ar.uniform_grid(30, 30)
    P.set_boundary_data(X=ar.cavity_boundary())
    P.set_eq_training_data(X=(x, y))
    P.log_every = 0
    P.save_every = 0
    P.train(num_epoch=steps, lr=1e-3)
    P.train(num_epoch=steps, lr=2e-4)        # second stage: new lr -> a second captured graph
    torch.cuda.synchronize()
    return P.engine.net.params.cpu().numpy().copy(), P.engine.net.adam_t, int(P.engine.net.adam_t_dev[0].item())


def test_graph_replay_is_bit_identical_to_eager(monkeypatch, tmp_path):
    monkeypatch.chdir(tmp_path)
    p_eager, t_eager, td_eager = _run_nsfnet(monkeypatch, False, 7)
    p_graph, t_graph, td_graph = _run_nsfnet(monkeypatch, True, 7)
    assert t_eager == t_graph == td_eager == td_graph == 14
    assert np.array_equal(p_eager, p_graph)


def test_graph_step_follows_oracle_adam_trajectory(monkeypatch):
    """Replayed steps against the fp64 oracle: bias corrections must advance inside the graph."""
    from nsfnet_amd import engine as eng
    from oracle import autograd_ref as ar, fwdmode_ref as fr
    monkeypatch.setenv("NSFNET_GRAPH", "1")
    dev = torch.device("cuda:0")
    L, H, Re = 2, 16, 100.0
    flat = ar.flat_params(ar.seeded_net(3, L, H, seed=5)).numpy().copy()
    E = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0)
    E.net.set_flat(torch.tensor(flat))
    x, y = (a.reshape(-1).astype(np.float32) for a in ar.uniform_grid(12, 12))
    xb, yb, ub, vb = (a.reshape(-1)[::16].astype(np.float32) for a in ar.cavity_boundary())
    E.set_collocation(x, y)
    E.set_boundary(xb, yb, ub, vb)
    p = flat.astype(np.float64); m = np.zeros_like(p); v = np.zeros_like(p)
    for t in range(1, 6):
        P = fr.unflatten(p, 2, 3, L, H)
        r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), Re, alpha_e=1.0)
        b = fr.bc_loss_and_grad(P, xb.astype(np.float64), yb.astype(np.float64), ub.astype(np.float64),
                                vb.astype(np.float64), alpha_b=10.0)
        p, m, v = fr.adam_step(p, r["grad"] + b["grad"], m, v, t, 1e-3)
        E.step(1e-3)
    torch.cuda.synchronize()
    assert len(E._graphs) == 1
    mine = E.net.params.cpu().numpy().astype(np.float64)
    upd = mine - flat
    assert np.linalg.norm(upd - (p - flat)) / np.linalg.norm(p - flat) < 2e-3


def test_init_vis_t_between_replays_matches_eager(monkeypatch):
    """PINN.init_vis_t() (ev-NSFnet/pinn_solver.py:138-140) after a step has been captured: the graph keeps
    reading and writing the lagged-viscosity allocation it was captured with, so the re-initialisation must
    land IN that allocation (or drop the graph) - bit-equal to the eager sequence either way."""
    from nsfnet_amd import engine as eng
    from oracle import autograd_ref as ar

    def run(graph):
        monkeypatch.setenv("NSFNET_GRAPH", "1" if graph else "0")
        dev = torch.device("cuda:0")
        E = eng.PinnEngine(dev, 3, 24, 1000.0, alpha_b=10.0, alpha_e=1.0, flavour="ev", n_hidden_e=2, hidden_e=12,
                           alpha_evm=0.05)
        E.net.set_flat(ar.flat_params(ar.seeded_net(3, 3, 24, seed=8)))
        E.net_e.set_flat(ar.flat_params(ar.seeded_net(1, 2, 12, seed=9)))
        x, y = (a.reshape(-1).astype(np.float32) for a in ar.uniform_grid(20, 20))
        xb, yb, ub, vb = (a.reshape(-1)[::16].astype(np.float32) for a in ar.cavity_boundary())
        E.set_collocation(x, y)
        E.set_boundary(xb, yb, ub, vb)
        ptr0 = E.plan_f.vis_t_minus.data_ptr()
        for _ in range(3):
            E.step(1e-3)
        E.alpha_evm = 0.02          # set_alpha_evm at a stage boundary, then the public re-initialisation
        E.init_vis_t()
        assert (not graph) or E.plan_f.vis_t_minus.data_ptr() == ptr0 or len(E._graphs) == 0
        for _ in range(3):
            E.step(1e-3)
        # supervised data dropped after capture: the captured steps must be dropped as well
        E.set_supervised(None, None, None, None)
        E.step(1e-3)
        torch.cuda.synchronize()
        return E.net.params.cpu().numpy().copy(), E.plan_f.vis_t.cpu().numpy().copy()

    p_e, v_e = run(False)
    p_g, v_g = run(True)
    assert np.array_equal(p_e, p_g) and np.array_equal(v_e, v_g)
