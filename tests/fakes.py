"""CPU stand-ins for the device objects of nsfnet_amd.engine (TEST INFRASTRUCTURE).

They let the HOST logic that surrounds the HIP calls - point sharding, the single
all-reduce of [grads | loss sums], global-count normalisation, Adam bookkeeping, the
solver classes' schedules - run under `gloo` on a machine without a GPU.  The arithmetic
inside the fakes is the fp64 oracle (oracle/fwdmode_ref.py); nothing here is reachable
from the product path.
"""
import numpy as np
import torch

from nsfnet_amd import engine as eng
from oracle import fwdmode_ref as fr


class FakeDeviceNet(eng.DeviceNet):
    def __init__(self, n_out, n_hidden, hidden, device, precision=None):
        self.lib = None
        self.handle = None
        self.n_out, self.n_hidden, self.hidden, self.device = n_out, n_hidden, hidden, torch.device("cpu")
        self.num_params = fr.param_count(2, n_out, n_hidden, hidden)
        self.params = torch.zeros(self.num_params, dtype=torch.float32)
        self.prep = torch.zeros(1)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.adam_t = 0
        self.adam_t_dev = torch.zeros(2, dtype=torch.int64)

    def __del__(self):
        pass

    def prepare(self):
        pass

    def pairs(self):
        return fr.unflatten(self.params.numpy().astype(np.float64), 2, self.n_out, self.n_hidden, self.hidden)

    def adam_step(self, grads, lr, betas=(0.9, 0.999), eps=1e-8):
        self.adam_t += 1
        p, m, v = fr.adam_step(self.params.numpy().astype(np.float64), grads.numpy().astype(np.float64),
                               self.m.numpy().astype(np.float64), self.v.numpy().astype(np.float64),
                               self.adam_t, lr, betas[0], betas[1], eps)
        self.params.copy_(torch.tensor(p, dtype=torch.float32))
        self.m.copy_(torch.tensor(m, dtype=torch.float32))
        self.v.copy_(torch.tensor(v, dtype=torch.float32))


def _vec(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float32).reshape(-1)).contiguous()


class FakeResidualPlan:
    def __init__(self, net, x, y, weights=None, with_backward=True, ws=None):
        self.net, self.x, self.y = net, _vec(x), _vec(y)
        self.ws = ws if ws is not None else torch.zeros(1)
        self.n = self.x.numel()
        self.npad = (self.n + 31) // 32 * 32
        self.fields = torch.zeros(eng.FLD_COUNT, self.npad)
        self.w = None if weights is None else _vec(weights)
        self.vis_t = torch.zeros(self.n)
        self.vis_t_minus = None
        self.ebar = None
        self.sums = torch.zeros(eng.NLOSS)
        self._grad = None

    def _xy(self):
        return self.x.numpy().astype(np.float64), self.y.numpy().astype(np.float64)

    def forward(self, Re, e=None, vis_t0=0.0, alpha_evm=0.0, scale=1.0, save=True, sums_out=None):
        x, y = self._xy()
        out, saved = fr.forward4(self.net.pairs(), x, y)
        ev = None if e is None else e.numpy().astype(np.float64)
        vt = np.zeros(self.n)
        if self.vis_t_minus is not None:
            vt = np.minimum(np.float32(vis_t0), self.vis_t_minus.numpy()).astype(np.float64)
            self.vis_t_minus = torch.tensor(alpha_evm * np.abs(ev), dtype=torch.float32)
        self.vis_t.copy_(torch.tensor(vt, dtype=torch.float32))
        eqs = fr.residuals(out, Re, vt, ev, scale)
        w = np.ones(self.n) if self.w is None else self.w.numpy().astype(np.float64)
        f = self.fields
        f.zero_()
        for name, val in (("u", out[:, 0, 0]), ("v", out[:, 1, 0]), ("p", out[:, 2, 0]),
                          ("u_x", out[:, 0, 1] * scale), ("u_y", out[:, 0, 2] * scale),
                          ("v_x", out[:, 1, 1] * scale), ("v_y", out[:, 1, 2] * scale)):
            f[eng.FLD[name], :self.n] = torch.tensor(val, dtype=torch.float32)
        so = self.sums if sums_out is None else sums_out
        so.zero_()
        for k, q in enumerate(eqs):
            f[eng.FLD["eq%d" % (k + 1)], :self.n] = torch.tensor(q, dtype=torch.float32)
            so[k] = float(np.sum(w * q * q))
        self._ctx = (Re, ev, vt, scale)

    def backward(self, Re, coef_eq, e=None, scale=1.0, want_ebar=False, phases=3):
        x, y = self._xy()
        Re, ev, vt, scale = self._ctx
        w = None if self.w is None else self.w.numpy().astype(np.float64)
        r = fr.pde_loss_and_grad(self.net.pairs(), x, y, Re, vis_t=vt, e=ev, w=w, scale=scale, coef_eq=list(coef_eq))
        self._grad = r["grad"]
        if want_ebar:
            self.ebar = torch.zeros((self.n + 127) // 128 * 128)
            self.ebar[:self.n] = torch.tensor(r["e_adj"], dtype=torch.float32)

    def field(self, name):
        return self.fields[eng.FLD[name], :self.n]


class FakeValuePlan:
    def __init__(self, net, x, y, targets=None, with_backward=True):
        self.net, self.x, self.y = net, _vec(x), _vec(y)
        self.n = self.x.numel()
        self.npad = (self.n + 127) // 128 * 128
        self.pred = torch.zeros(net.n_out, self.n)
        self.targets = [None, None, None]
        if targets is not None:
            for c, t in enumerate(targets):
                if t is not None:
                    self.targets[c] = _vec(t)
        self.sums = torch.zeros(eng.NLOSS)
        self._grad = None

    def forward(self, coef=(0.0, 0.0, 0.0), save=False, use_targets=True, sums_out=None):
        x, y = self.x.numpy().astype(np.float64), self.y.numpy().astype(np.float64)
        out, self._saved = fr.forward1(self.net.pairs(), x, y)
        self.pred.copy_(torch.tensor(out.T, dtype=torch.float32))
        so = self.sums if sums_out is None else sums_out
        so.zero_()
        adj = np.zeros_like(out)
        for c in range(self.net.n_out):
            t = self.targets[c]
            if use_targets and t is not None:
                tv = t.numpy().astype(np.float64)
                ok = np.isfinite(tv)
                d = np.where(ok, out[:, c] - np.where(ok, tv, 0.0), 0.0)
                so[c] = float(np.sum(d * d))
                if c == 2:
                    so[3] = float(ok.sum())
                adj[:, c] = coef[c] * d
        self._adj = adj

    def backward(self, out_adj=None):
        x, y = self.x.numpy().astype(np.float64), self.y.numpy().astype(np.float64)
        adj = self._adj if out_adj is None else out_adj.numpy().astype(np.float64)[: self.n].reshape(-1, 1)
        self._grad = fr.backward1(self.net.pairs(), x, y, self._saved, adj)


def fake_grad_reduce(net, plans, grads_out, accumulate=False):
    g = sum(p._grad for p in plans)
    t = torch.tensor(g, dtype=torch.float32)
    if accumulate:
        grads_out.add_(t)
    else:
        grads_out.copy_(t)


def install(monkeypatch=None):
    """Swap the device classes of nsfnet_amd.engine (and the solvers' device probe) for the fakes."""
    from nsfnet_amd import pinn_solver as ps, ev_pinn_solver as es
    repl = [(eng, "DeviceNet", FakeDeviceNet), (eng, "ResidualPlan", FakeResidualPlan),
            (eng, "ValuePlan", FakeValuePlan), (eng, "grad_reduce", fake_grad_reduce),
            (ps, "default_device", lambda: torch.device("cpu")), (es, "default_device", lambda: torch.device("cpu")),
            (torch.cuda, "set_device", lambda *_a, **_k: None)]
    for mod, name, val in repl:
        if monkeypatch is not None:
            monkeypatch.setattr(mod, name, val)
        else:
            setattr(mod, name, val)
