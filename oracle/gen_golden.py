#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

TEST INFRASTRUCTURE.  Runs only in the build container, where the upstream
checkout is mounted read-only at /root/reference; the reference never travels
to the GPU box - only the small .npz vectors written here do (inputs and
expected outputs, no reference source).  Re-run with

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden.py

The reference sets no seeds and samples points from the unseeded global numpy
RNG (tools.py:42-43), so every case injects seeded weights and points.
"""
import os
import sys
import importlib
import tempfile

import numpy as np
import torch

REF = os.environ.get("NSFNET_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _import_flavour(sub):
    """Import <REF>/<sub>/{net,tools,cavity_data,pinn_solver}.py under fresh names."""
    for m in ("net", "tools", "cavity_data", "pinn_solver"):
        sys.modules.pop(m, None)
    sys.path.insert(0, os.path.join(REF, sub))
    try:
        mods = {m: importlib.import_module(m) for m in ("net", "cavity_data", "pinn_solver")}
    finally:
        sys.path.pop(0)
    for m in ("net", "tools", "cavity_data", "pinn_solver"):
        sys.modules.pop(m, None)
    return mods


def _flat(net, grad=False):
    ps = list(net.parameters())
    if grad:
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps]).numpy().copy()
    return torch.cat([p.detach().reshape(-1) for p in ps]).numpy().copy()


def _points(n, seed):
    rng = np.random.RandomState(seed)
    return rng.rand(n, 1), rng.rand(n, 1)


def gen_nsfnet(mods, name, L, H, N, Re, seed, steps, alpha_b=10.0, alpha_e=1.0, lr=1e-3,
               store_weights=True, stride=1):
    ps = mods["pinn_solver"]
    torch.manual_seed(seed)
    P = ps.PysicsInformedNeuralNetwork(Re=Re, layers=L, hidden_size=H, N_f=N,
                                       bc_weight=alpha_b, eq_weight=alpha_e,
                                       learning_rate=lr, num_ins=2, num_outs=3)
    bc = mods["cavity_data"].DataLoader(N_f=N).loading_boundary_data()
    x, y = _points(N, seed + 1)
    P.set_boundary_data(X=bc)
    P.set_eq_training_data(X=(x, y))
    w0 = _flat(P.net)
    rec = dict(L=L, H=H, N=N, Re=Re, seed=seed, alpha_b=alpha_b, alpha_e=alpha_e, lr=lr,
               x=x, y=y, stride=stride,
               w0_sample=w0[::stride], w0_sum=np.float64(w0.astype(np.float64).sum()))
    if store_weights:
        rec["w0"] = w0
    if name.startswith("nsfnet_4x50"):
        rec.update(x_b=bc[0], y_b=bc[1], u_b=bc[2], v_b=bc[3])
    losses, params = [], []
    for k in range(steps):
        loss, (loss_e, loss_b) = P.fwd_computing_loss_2d()
        if k == 0:
            rec.update(eq1=P.eq1_pred.detach().numpy().copy(), eq2=P.eq2_pred.detach().numpy().copy(),
                       eq3=P.eq3_pred.detach().numpy().copy(),
                       u_pred_b=P.u_pred_b.detach().numpy().copy(),
                       v_pred_b=P.v_pred_b.detach().numpy().copy())
        loss.backward()
        if k == 0:
            rec["grad0"] = _flat(P.net, grad=True)[::stride]
        losses.append([float(loss), float(loss_b), float(P.loss_eq1), float(P.loss_eq2), float(P.loss_eq3)])
        P.opt.step()
        P.opt.zero_grad()
        params.append(_flat(P.net)[::stride])
    rec["losses"] = np.array(losses, dtype=np.float64)
    rec["params_after"] = np.stack(params)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "loss0", losses[0])


def gen_nsfnet_l2(mods, name, L, H, N, Re, seed, steps, alpha_b=10.0, alpha_e=1.0, lr=1e-3):
    """loss_mode='L2' of the plain solver (NSFnet/pinn_solver.py:202-204, 214-217): 2-norms instead of mean squares.
    No script of the reference selects it; the branch is driven here directly."""
    ps = mods["pinn_solver"]
    torch.manual_seed(seed)
    P = ps.PysicsInformedNeuralNetwork(Re=Re, layers=L, hidden_size=H, N_f=N, bc_weight=alpha_b, eq_weight=alpha_e,
                                       learning_rate=lr, num_ins=2, num_outs=3)
    bc = mods["cavity_data"].DataLoader(N_f=N).loading_boundary_data()
    bc = tuple(a[::8] for a in bc)
    x, y = _points(N, seed + 1)
    P.set_boundary_data(X=bc)
    P.set_eq_training_data(X=(x, y))
    rec = dict(L=L, H=H, N=N, Re=Re, seed=seed, alpha_b=alpha_b, alpha_e=alpha_e, lr=lr, x=x, y=y, w0=_flat(P.net),
               x_b=bc[0], y_b=bc[1], u_b=bc[2], v_b=bc[3])
    losses, params = [], []
    for k in range(steps):
        loss, (loss_e, loss_b) = P.fwd_computing_loss_2d(loss_mode='L2')
        loss.backward()
        if k == 0:
            rec["grad0"] = _flat(P.net, grad=True)
        losses.append([float(loss), float(loss_b), float(loss_e)])
        P.opt.step()
        P.opt.zero_grad()
        params.append(_flat(P.net))
    rec["losses"] = np.array(losses, dtype=np.float64)
    rec["params_after"] = np.stack(params)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "loss0", losses[0])


def _make_ev(mods, L, H, L1, H1, Re, alpha_evm, alpha_b, alpha_e, lr, seed, coord_scale=1.0):
    """ev ctor hard-requires a GPU (ev-NSFnet/pinn_solver.py:62-63); build the
    object by hand with the attributes the ctor would set."""
    ps, net = mods["pinn_solver"], mods["net"]
    C = ps.PysicsInformedNeuralNetwork
    P = object.__new__(C)
    P.rank = P.local_rank = 0
    P.world_size = 1
    P.device = torch.device("cpu")
    P.is_distributed = False
    P.evm = None
    P.Re = Re
    P.vis_t0 = 20.0 / Re
    P.layers, P.layers_1, P.hidden_size, P.hidden_size_1, P.N_f = L, L1, H, H1, 0
    P.current_stage = " "
    P.alpha_evm, P.alpha_b, P.alpha_e, P.alpha_s = alpha_evm, alpha_b, alpha_e, 0.0
    P.loss_i = P.loss_o = P.loss_b = P.loss_e = P.loss_s = 0.0
    P.x_s = P.y_s = P.u_s = P.v_s = P.p_s = None
    P._p_mask = None
    P.supervision_point_count = P.supervision_total_points = 0
    P.supervision_has_data = P.supervision_enabled = False
    P.eq_weights = None
    P.coord_scale, P.coord_scale_sq = coord_scale, coord_scale ** 2
    P.vis_t = P.vis_t_minus = None
    torch.manual_seed(seed)
    P.net = net.FCNet(2, 3, L, H, torch.nn.Tanh)
    P.net_1 = net.FCNet(2, 1, L1, H1, torch.nn.Tanh)
    P.opt = torch.optim.Adam(list(P.net.parameters()) + list(P.net_1.parameters()), lr=lr, weight_decay=0.0)
    P.print_log = lambda *a, **k: None
    P.save = lambda *a, **k: None
    return P


def gen_ev(mods, name, L, H, L1, H1, N, Re, alpha_evm, seed, steps, alpha_b=10.0, alpha_e=1.0,
           lr=1e-3, sdf=False, coord_scale=1.0, nb_stride=4):
    P = _make_ev(mods, L, H, L1, H1, Re, alpha_evm, alpha_b, alpha_e, lr, seed, coord_scale)
    dl = mods["cavity_data"].DataLoader(N_f=N, N_b=1000)
    bc = dl.loading_boundary_data()
    bc = tuple(a[::nb_stride] for a in bc)
    x, y = _points(N, seed + 1)
    if coord_scale != 1.0:   # cavity_data.py:135-136 maps [0,1] -> [-1,1]
        x, y = x * 2.0 - 1.0, y * 2.0 - 1.0
        bc = (bc[0] * 2.0 - 1.0, bc[1] * 2.0 - 1.0, bc[2], bc[3])
    w = None
    if sdf:
        rng = np.random.RandomState(seed + 2)
        w = (0.2 + rng.rand(N)).astype(np.float32)
        w = w / w.mean()
    P.set_boundary_data(X=bc)
    P.set_eq_training_data(X=(x, y), weights=w)
    rec = dict(L=L, H=H, L1=L1, H1=H1, N=N, Re=Re, alpha_evm=alpha_evm, seed=seed,
               alpha_b=alpha_b, alpha_e=alpha_e, lr=lr, coord_scale=coord_scale,
               x=x, y=y, x_b=bc[0], y_b=bc[1], u_b=bc[2], v_b=bc[3],
               w0=_flat(P.net), w0_e=_flat(P.net_1), vis_t_minus0=np.asarray(P.vis_t_minus).copy())
    if w is not None:
        rec["weights"] = w
    # drive the reference's own solve_Adam (pinn_solver.py:440-487) and observe it through
    # the loss_func it is handed: call k sees the parameters/gradients left by step k-1.
    losses, vis, params = [], [], []
    real = P.fwd_computing_loss_2d
    calls = {"n": 0}

    def spy():
        k = calls["n"]
        calls["n"] += 1
        if k >= 1:
            params.append(_flat(P.net))
            if k == 1:
                rec["grad0"] = _flat(P.net, grad=True)
        if k == steps:          # one extra call only to observe the last step; abort the loop
            raise StopIteration
        out = real()
        if k == 0:
            for i, q in enumerate((P.eq1_pred, P.eq2_pred, P.eq3_pred, P.eq4_pred)):
                rec["eq%d" % (i + 1)] = q.detach().numpy().copy()
        losses.append([float(out[0]), float(P.loss_b), float(P.loss_eq1), float(P.loss_eq2),
                       float(P.loss_eq3), float(P.loss_eq4)])
        vis.append(P.vis_t.detach().numpy().reshape(-1).copy())
        return out

    try:
        P.solve_Adam(spy, num_epoch=steps + 1)
    except StopIteration:
        pass
    rec.update(losses=np.array(losses), vis_t=np.stack(vis), params_after=np.stack(params),
               params_e_after=_flat(P.net_1))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "loss0", losses[0])


def gen_ev_supervised(mods, name, L=3, H=24, L1=2, H1=10, N=160, NS=37, Re=2000, alpha_evm=0.04, alpha_s=0.7,
                      seed=31, steps=3, alpha_b=10.0, alpha_e=1.0, lr=1e-3, p_mode="nan"):
    """Supervised-data branch of the reference loss (ev-NSFnet/pinn_solver.py:202-251 set_supervised_data,
    :399-411 loss_s): per-output means, pressure averaged over its FINITE targets only.  p_mode: "nan" = some
    NaN pressure targets, "none" = no pressure array, "allnan" = every pressure target masked."""
    P = _make_ev(mods, L, H, L1, H1, Re, alpha_evm, alpha_b, alpha_e, lr, seed)
    P.alpha_s = alpha_s
    dl = mods["cavity_data"].DataLoader(N_f=N, N_b=1000)
    bc = tuple(a[::16] for a in dl.loading_boundary_data())
    x, y = _points(N, seed + 1)
    rng = np.random.RandomState(seed + 3)
    xs, ys = rng.rand(NS, 1), rng.rand(NS, 1)
    us, vs = rng.randn(NS, 1) * 0.3, rng.randn(NS, 1) * 0.2
    ps_ = rng.randn(NS, 1) * 0.1
    if p_mode == "nan":
        ps_[rng.rand(NS) < 0.35] = np.nan
    elif p_mode == "allnan":
        ps_[:] = np.nan
    elif p_mode == "none":
        ps_ = None
    P.set_boundary_data(X=bc)
    P.set_eq_training_data(X=(x, y))
    P.set_supervised_data((xs, ys, us, vs, ps_))
    assert P.supervision_enabled
    rec = dict(L=L, H=H, L1=L1, H1=H1, N=N, Re=Re, alpha_evm=alpha_evm, seed=seed, alpha_s=alpha_s,
               alpha_b=alpha_b, alpha_e=alpha_e, lr=lr, x=x, y=y, x_b=bc[0], y_b=bc[1], u_b=bc[2], v_b=bc[3],
               x_s=xs, y_s=ys, u_s=us, v_s=vs, w0=_flat(P.net), w0_e=_flat(P.net_1))
    if ps_ is not None:
        rec["p_s"] = ps_
    losses, params = [], []
    real = P.fwd_computing_loss_2d
    calls = {"n": 0}

    def spy():
        k = calls["n"]
        calls["n"] += 1
        if k >= 1:
            params.append(_flat(P.net))
            if k == 1:
                rec["grad0"] = _flat(P.net, grad=True)
        if k == steps:
            raise StopIteration
        out = real()
        losses.append([float(out[0]), float(P.loss_b), float(P.loss_e), float(P.loss_s)])
        return out

    try:
        P.solve_Adam(spy, num_epoch=steps + 1)
    except StopIteration:
        pass
    # the same first step with the supervised weight switched off (set_supervised_loss_weight(0), :253-255)
    P2 = _make_ev(mods, L, H, L1, H1, Re, alpha_evm, alpha_b, alpha_e, lr, seed)
    P2.alpha_s = alpha_s
    P2.set_boundary_data(X=bc); P2.set_eq_training_data(X=(x, y)); P2.set_supervised_data((xs, ys, us, vs, ps_))
    P2.set_supervised_loss_weight(0.0)
    out = P2.fwd_computing_loss_2d()
    rec["loss_alpha0"] = np.array([float(out[0]), float(P2.loss_s)])
    rec.update(losses=np.array(losses), params_after=np.stack(params))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "losses0 [total, b, e, s]", losses[0])


def gen_ev_freeze(mods, name, seed=7):
    """Steps 10000..10002 of solve_Adam on a tiny net: net_1 trains for exactly
    one step and Adam is re-created twice (pinn_solver.py:459-462, 489-511)."""
    L, H, L1, H1, N, Re = 2, 8, 2, 6, 48, 1000
    P = _make_ev(mods, L, H, L1, H1, Re, 0.05, 10.0, 1.0, 1e-3, seed)
    bc = mods["cavity_data"].DataLoader(N_f=N, N_b=1000).loading_boundary_data()
    bc = tuple(a[::64] for a in bc)
    x, y = _points(N, seed + 1)
    P.set_boundary_data(X=bc)
    P.set_eq_training_data(X=(x, y))
    snaps = {}
    calls = {"n": 0}
    real = P.fwd_computing_loss_2d

    def spy():
        k = calls["n"]
        if k in (10000, 10001, 10002, 10003):
            snaps["p_%d" % k] = _flat(P.net)
            snaps["pe_%d" % k] = _flat(P.net_1)
            snaps["vtm_%d" % k] = np.asarray(P.vis_t_minus).copy()
        calls["n"] += 1
        out = real()
        if k in (10000, 10001, 10002):
            snaps["loss_%d" % k] = np.float64(float(out[0]))
        return out

    P.solve_Adam(spy, num_epoch=10004)
    rec = dict(L=L, H=H, L1=L1, H1=H1, N=N, Re=Re, alpha_evm=0.05, alpha_b=10.0, alpha_e=1.0, lr=1e-3,
               x=x, y=y, x_b=bc[0], y_b=bc[1], u_b=bc[2], v_b=bc[3], **snaps)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "done;  d(pe) 10000->10001 =", np.abs(snaps["pe_10001"] - snaps["pe_10000"]).max(),
          " 10001->10002 =", np.abs(snaps["pe_10002"] - snaps["pe_10001"]).max())


def gen_data_prep(ns, ev):
    """Reference sampler under a seeded global RNG (tools.py:30-83) and SDF weights
    (ev-NSFnet/cavity_data.py:118-130)."""
    import types
    tools = sys.modules.get("tools")
    dl = ns["cavity_data"].DataLoader(N_f=600)
    bc = dl.loading_boundary_data()
    np.random.seed(123)
    x, y = dl.loading_training_data()           # LHSample + sort_pts, reference code
    np.random.seed(321)
    lhs = ns["cavity_data"].LHSample(2, [[0.0, 1.0], [-1.0, 1.0]], 257)
    cfg = types.SimpleNamespace(enabled=True, min_weight=0.3, decay=4.0)
    dle = ev["cavity_data"].DataLoader(N_f=400, sort_training_points=False, sdf_weighting=cfg, coord_transform=True)
    dle.loading_boundary_data()
    np.random.seed(77)
    xe, ye = dle.loading_training_data()
    np.savez_compressed(os.path.join(OUT, "data_prep.npz"), x_sorted=x, y_sorted=y, lhs=lhs,
                        xe=xe, ye=ye, sdf=dle.get_sdf_weights(), coord_scale=dle.get_coord_scale())
    print("data_prep done", x.shape, lhs.shape, xe.shape)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    only = sys.argv[1] if len(sys.argv) > 1 else None      # `gen_golden.py l2`: only the fixture added in round 3
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        ns = _import_flavour("NSFnet")
        if only == "l2":
            gen_nsfnet_l2(ns, "nsfnet_l2_3x24_re400", 3, 24, 200, 400, 5, 3)
            os.chdir("/tmp")
            return
        gen_nsfnet(ns, "nsfnet_4x50_re100", 4, 50, 2048, 100, 0, 3)
        gen_nsfnet(ns, "nsfnet_2x16_re1000", 2, 16, 300, 1000, 3, 5)
        gen_nsfnet(ns, "nsfnet_6x256_re2000_n256", 6, 256, 256, 2000, 1234, 1, store_weights=False, stride=16)
        if only in (None, "l2"):
            gen_nsfnet_l2(ns, "nsfnet_l2_3x24_re400", 3, 24, 200, 400, 5, 3)
        ev = _import_flavour("ev-NSFnet")
        gen_ev(ev, "ev_4x50_4x40_re4000", 4, 50, 4, 40, 1024, 4000, 0.05, 11, 4)
        gen_ev(ev, "ev_2x16_sdf_scaled", 2, 16, 2, 12, 256, 3000, 0.03, 21, 4, sdf=True, coord_scale=2.0)
        gen_ev_freeze(ev, "ev_freeze_2x8")
        gen_ev_supervised(ev, "ev_sup_3x24_nanp", p_mode="nan")
        gen_ev_supervised(ev, "ev_sup_3x24_nop", p_mode="none", seed=41)
        gen_ev_supervised(ev, "ev_sup_3x24_allnanp", p_mode="allnan", seed=43, steps=1)
        gen_data_prep(ns, ev)
        os.chdir("/tmp")


if __name__ == "__main__":
    main()
