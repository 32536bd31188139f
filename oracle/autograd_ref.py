"""Torch-autograd restatement of the reference PINN step (TEST INFRASTRUCTURE).

Every function cites the reference lines it follows (paths relative to the
upstream repo latteine1217/NSFnet).  This module is the CPU oracle and the
``cpu_baseline`` of bench.py; it is never imported by ``nsfnet_amd``.

The algorithm: an MLP (u,v,p)=net(x,y); nine reverse-mode sweeps
(``torch.autograd.grad(..., create_graph=True)``) give u_x,u_y,u_xx,u_yy,
v_x,v_y,v_xx,v_yy,p_x,p_y; momentum/continuity (and entropy) residuals are
squared-mean'ed into the loss; ``loss.backward()`` differentiates through all
of it; Adam updates the parameters.
"""
from collections import OrderedDict

import numpy as np
import torch


# --------------------------------------------------------------------------
# model: NSFnet/net.py:22-54 (identical file in ev-NSFnet/)
# --------------------------------------------------------------------------
class RefFCNet(torch.nn.Module):
    """[n_in] + [hidden]*n_hidden + [n_out] tanh MLP.

    net.py:30-46: ``num_layers`` counts HIDDEN layers, so there are
    num_layers+1 Linear modules named layer_0..layer_{num_layers}; Tanh
    (activation_i) follows every Linear but the last.  state_dict keys are
    ``layers.layer_{i}.weight|bias`` because the Sequential is stored on the
    attribute ``layers`` (net.py:50).
    """

    def __init__(self, n_in=2, n_out=3, n_hidden=4, hidden=50):
        super().__init__()
        widths = [n_in] + [hidden] * n_hidden + [n_out]
        mods = OrderedDict()
        last = len(widths) - 2
        for i in range(len(widths) - 1):
            mods["layer_%d" % i] = torch.nn.Linear(widths[i], widths[i + 1])
            if i != last:
                mods["activation_%d" % i] = torch.nn.Tanh()
        self.layers = torch.nn.Sequential(mods)

    def forward(self, X):
        return self.layers(X)


def flat_params(net):
    """state_dict-order flat fp32/fp64 vector (weight, bias per layer)."""
    return torch.cat([p.detach().reshape(-1) for p in net.parameters()])


def flat_grads(net):
    return torch.cat([
        (p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
        for p in net.parameters()])


def _grad_sum(out, wrt):
    """NSFnet/pinn_solver.py:165-186, ev-NSFnet/pinn_solver.py:344-361:
    d(sum(out))/d(wrt_i) with create_graph=True, allow_unused=True and
    None replaced by zeros."""
    gs = torch.autograd.grad([out], wrt, grad_outputs=[torch.ones_like(out)],
                             create_graph=True, allow_unused=True)
    return [g if g is not None else torch.zeros_like(w) for g, w in zip(gs, wrt)]


def ns_fields(net, x, y):
    """u,v,p and the ten derivatives the residuals consume.
    NSFnet/pinn_solver.py:132-148, ev-NSFnet/pinn_solver.py:290-309."""
    X = torch.cat((x, y), dim=1)
    out = net(X)
    u, v, p = out[:, 0:1], out[:, 1:2], out[:, 2:3]
    u_x, u_y = _grad_sum(u, [x, y])
    u_xx = _grad_sum(u_x, [x])[0]
    u_yy = _grad_sum(u_y, [y])[0]
    v_x, v_y = _grad_sum(v, [x, y])
    v_xx = _grad_sum(v_x, [x])[0]
    v_yy = _grad_sum(v_y, [y])[0]
    p_x, p_y = _grad_sum(p, [x, y])
    return dict(u=u, v=v, p=p, u_x=u_x, u_y=u_y, u_xx=u_xx, u_yy=u_yy,
                v_x=v_x, v_y=v_y, v_xx=v_xx, v_yy=v_yy, p_x=p_x, p_y=p_y)


def nsfnet_residuals(net, x, y, Re):
    """eq1..3 of plain NSFnet.  NSFnet/pinn_solver.py:159-163."""
    f = ns_fields(net, x, y)
    nu = 1.0 / Re
    eq1 = (f["u"] * f["u_x"] + f["v"] * f["u_y"]) + f["p_x"] - nu * (f["u_xx"] + f["u_yy"])
    eq2 = (f["u"] * f["v_x"] + f["v"] * f["v_y"]) + f["p_y"] - nu * (f["v_xx"] + f["v_yy"])
    eq3 = f["u_x"] + f["v_y"]
    return eq1, eq2, eq3, f


def bc_loss(net, x_b, y_b, u_b, v_b):
    """mean((u_b-u)^2)+mean((v_b-v)^2).  NSFnet/pinn_solver.py:199-207,
    ev-NSFnet/pinn_solver.py:374-379."""
    out = net(torch.cat((x_b, y_b), dim=1))
    return (torch.mean(torch.square(u_b.reshape(-1) - out[:, 0])) +
            torch.mean(torch.square(v_b.reshape(-1) - out[:, 1])))


class NSFnetOracle:
    """Plain NSFnet step.  Follows NSFnet/pinn_solver.py:197-278.

    Data are (N,1) tensors of ``dtype`` (reference: fp64 numpy -> .float(),
    pinn_solver.py:82-99).  Adam: torch defaults, weight_decay=0
    (pinn_solver.py:76-79); order loss -> backward -> step -> zero_grad
    (pinn_solver.py:251-254); optimizer state persists across train() calls
    (pinn_solver.py:228-238).
    """

    def __init__(self, net, Re, alpha_b=1.0, alpha_e=1.0, lr=1e-3, loss_mode="MSE"):
        self.net, self.Re = net, Re
        self.alpha_b, self.alpha_e = alpha_b, alpha_e
        self.loss_mode = loss_mode      # 'MSE' | 'L2' (NSFnet/pinn_solver.py:202-217; no script selects 'L2')
        self.opt = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=0)

    def set_data(self, x_f, y_f, x_b, y_b, u_b, v_b):
        dt = next(self.net.parameters()).dtype
        t = lambda a, rg=False: torch.tensor(np.asarray(a), dtype=dt).reshape(-1, 1).requires_grad_(rg)
        self.x_f, self.y_f = t(x_f, True), t(y_f, True)
        self.x_b, self.y_b, self.u_b, self.v_b = t(x_b), t(y_b), t(u_b), t(v_b)

    def loss(self):
        self.eq1, self.eq2, self.eq3, self.fields = nsfnet_residuals(
            self.net, self.x_f, self.y_f, self.Re)
        if self.loss_mode == "L2":      # 2-norms instead of mean squares (NSFnet/pinn_solver.py:202-204, 214-217)
            out = self.net(torch.cat((self.x_b, self.y_b), dim=1))
            self.loss_b = (torch.norm(self.u_b.reshape(-1) - out[:, 0], p=2) +
                           torch.norm(self.v_b.reshape(-1) - out[:, 1], p=2))
            self.loss_eq = [torch.norm(e.reshape(-1), p=2) for e in (self.eq1, self.eq2, self.eq3)]
        else:
            self.loss_b = bc_loss(self.net, self.x_b, self.y_b, self.u_b, self.v_b)
            self.loss_eq = [torch.mean(torch.square(e.reshape(-1))) for e in (self.eq1, self.eq2, self.eq3)]
        self.loss_e = self.loss_eq[0] + self.loss_eq[1] + self.loss_eq[2]
        self.total = self.alpha_b * self.loss_b + self.alpha_e * self.loss_e
        return self.total

    def step(self, lr=None):
        if lr is not None:
            self.opt.param_groups[0]["lr"] = lr
        total = self.loss()
        total.backward()
        self.grads = flat_grads(self.net).clone()
        self.opt.step()
        self.opt.zero_grad()
        return float(total.detach())


class EvNSFnetOracle:
    """ev-NSFnet step on one rank.  Follows ev-NSFnet/pinn_solver.py:138-140
    (init_vis_t), :290-342 (residuals incl. lagged artificial viscosity),
    :372-428 (loss), :440-511 (Adam loop, freeze schedule).

    vis_t  = min(20/Re, vis_t_minus)          (pinn_solver.py:67, :327-331)
    vis_t_minus <- alpha_evm*|e| (detached)   (pinn_solver.py:334)
    eq4    = eq1*(u-.5)+eq2*(v-.5)-e          (pinn_solver.py:341)
    loss_e = m(eq1)+m(eq2)+m(eq3)+0.1*m(eq4), m(r)=mean(w*r^2) (:387-397)
    loss_s = mean((u_s-u)^2)+mean((v_s-v)^2)+mean over FINITE p_s of (p_s-p)^2   (:399-411)
    loss   = alpha_b*loss_b + alpha_e*loss_e + alpha_s*loss_s                    (:426)
    """

    def __init__(self, net, net_e, Re, alpha_evm, alpha_b=10.0, alpha_e=1.0, lr=1e-3,
                 coord_scale=1.0, alpha_s=0.0):
        self.net, self.net_e, self.Re = net, net_e, Re
        self.vis_t0 = 20.0 / Re
        self.alpha_evm, self.alpha_b, self.alpha_e = alpha_evm, alpha_b, alpha_e
        self.alpha_s = alpha_s
        self.sup = None
        self.scale, self.scale_sq = float(coord_scale), float(coord_scale) ** 2
        self.w = None
        self.vis_t_minus = None
        self.lr = lr
        self.freeze_e()

    # ev-NSFnet/pinn_solver.py:489-511 : every (de)freeze re-creates Adam
    def freeze_e(self):
        for p in self.net_e.parameters():
            p.requires_grad = False
        self.opt = torch.optim.Adam(list(self.net.parameters()), lr=self.lr, weight_decay=0.0)

    def defreeze_e(self):
        for p in self.net_e.parameters():
            p.requires_grad = True
        self.opt = torch.optim.Adam(list(self.net.parameters()) + list(self.net_e.parameters()),
                                    lr=self.lr, weight_decay=0.0)

    def set_data(self, x_f, y_f, x_b, y_b, u_b, v_b, weights=None):
        dt = next(self.net.parameters()).dtype
        t = lambda a, rg=False: torch.tensor(np.asarray(a), dtype=dt).reshape(-1, 1).requires_grad_(rg)
        self.x_f, self.y_f = t(x_f, True), t(y_f, True)
        self.x_b, self.y_b, self.u_b, self.v_b = t(x_b), t(y_b), t(u_b), t(v_b)
        self.w = None if weights is None else torch.tensor(np.asarray(weights), dtype=dt).reshape(-1)
        # init_vis_t, pinn_solver.py:138-140,184
        with torch.no_grad():
            e = self.net_e(torch.cat((self.x_f, self.y_f), dim=1))[:, 0:1]
        self.vis_t_minus = self.alpha_evm * torch.abs(e).detach()

    def set_supervised(self, x_s, y_s, u_s, v_s, p_s=None):
        """ev-NSFnet/pinn_solver.py:202-251: fp32 copies of the samples; the pressure mask is
        np.isfinite of the targets (:247-249); None clears the data (:194-200)."""
        if x_s is None or len(np.asarray(x_s)) == 0:
            self.sup = None
            return
        dt = next(self.net.parameters()).dtype
        t = lambda a: torch.tensor(np.asarray(a), dtype=dt).reshape(-1, 1)
        mask = None if p_s is None else torch.tensor(np.isfinite(np.asarray(p_s)).reshape(-1))
        self.sup = (t(x_s), t(y_s), t(u_s), t(v_s), None if p_s is None else t(p_s), mask)

    def supervised_loss(self):
        """ev-NSFnet/pinn_solver.py:399-411 (enabled only with data and alpha_s != 0, :251, :255)."""
        if self.sup is None or self.alpha_s == 0.0:
            return torch.zeros((), dtype=next(self.net.parameters()).dtype)
        x_s, y_s, u_s, v_s, p_s, mask = self.sup
        out = self.net(torch.cat((x_s, y_s), dim=1))
        loss_u = torch.mean(torch.square(u_s.view(-1) - out[:, 0]))
        loss_v = torch.mean(torch.square(v_s.view(-1) - out[:, 1]))
        loss_p = torch.zeros((), dtype=out.dtype)
        if p_s is not None and mask is not None and bool(mask.any()):
            loss_p = torch.mean(torch.square(p_s.view(-1)[mask] - out[:, 2][mask]))
        return loss_u + loss_v + loss_p

    def loss(self):
        self.loss_b = bc_loss(self.net, self.x_b, self.y_b, self.u_b, self.v_b)
        x, y = self.x_f, self.y_f
        f = ns_fields(self.net, x, y)
        e = self.net_e(torch.cat((x, y), dim=1))[:, 0:1]
        s, s2 = self.scale, self.scale_sq
        u, v = f["u"], f["v"]
        u_x, u_y, v_x, v_y = f["u_x"] * s, f["u_y"] * s, f["v_x"] * s, f["v_y"] * s
        p_x, p_y = f["p_x"] * s, f["p_y"] * s
        lap_u = f["u_xx"] * s2 + f["u_yy"] * s2
        lap_v = f["v_xx"] * s2 + f["v_yy"] * s2
        self.vis_t = torch.minimum(torch.full_like(self.vis_t_minus, self.vis_t0), self.vis_t_minus)
        self.vis_t_minus = self.alpha_evm * torch.abs(e).detach()
        nu = 1.0 / self.Re + self.vis_t
        eq1 = (u * u_x + v * u_y) + p_x - nu * lap_u
        eq2 = (u * v_x + v * v_y) + p_y - nu * lap_v
        eq3 = u_x + v_y
        eq4 = (eq1 * (u - 0.5) + eq2 * (v - 0.5)) - e
        self.eq = (eq1, eq2, eq3, eq4)
        self.fields, self.e = f, e

        def wmse(r):
            r = r.reshape(-1)
            if self.w is not None:
                r = r * torch.sqrt(self.w)
            return torch.mean(torch.square(r))

        self.loss_eq = [wmse(r) for r in self.eq]
        self.loss_e = self.loss_eq[0] + self.loss_eq[1] + self.loss_eq[2] + 0.1 * self.loss_eq[3]
        self.loss_s = self.supervised_loss()
        self.total = self.alpha_b * self.loss_b + self.alpha_e * self.loss_e + self.alpha_s * self.loss_s
        return self.total

    def step(self, epoch_id=None):
        """One iteration of solve_Adam's loop body (pinn_solver.py:456-472)."""
        if epoch_id is not None:
            if epoch_id != 0 and epoch_id % 10000 == 0:
                self.defreeze_e()
            if (epoch_id - 1) % 10000 == 0:
                self.freeze_e()
        total = self.loss()
        self.opt.zero_grad()
        total.backward()
        self.grads = flat_grads(self.net).clone()
        self.grads_e = flat_grads(self.net_e).clone()
        self.opt.step()
        return float(total.detach())


# --------------------------------------------------------------------------
# data helpers shared by tests / bench
# --------------------------------------------------------------------------
def cavity_boundary(nx=513):
    """The reference's 4*513 boundary set with the regularised lid
    u=1-cosh(10(x-.5))/cosh(5).  NSFnet/cavity_data.py:38-63."""
    s = np.linspace(0.0, 1.0, nx)
    lid = 1.0 - np.cosh(10.0 * (s - 0.5)) / np.cosh(5.0)
    zeros, ones = np.zeros(nx), np.ones(nx)
    x_b = np.concatenate([s, s, zeros, ones])
    y_b = np.concatenate([zeros, ones, s, s])
    u_b = np.concatenate([zeros, lid, zeros, zeros])
    v_b = np.zeros(4 * nx)
    return tuple(a.reshape(-1, 1) for a in (x_b, y_b, u_b, v_b))


def uniform_grid(nx, ny):
    """Cell-centred uniform grid on (0,1)^2, row-major."""
    xs = (np.arange(nx) + 0.5) / nx
    ys = (np.arange(ny) + 0.5) / ny
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    return X.reshape(-1, 1), Y.reshape(-1, 1)


def seeded_net(n_out, n_hidden, hidden, seed=1234, dtype=torch.float32):
    torch.manual_seed(seed)
    return RefFCNet(2, n_out, n_hidden, hidden).to(dtype)
