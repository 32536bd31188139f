"""Host-side driver of the HIP PINN pipeline (one process = one GPU).

PyTorch is plumbing here: it owns device memory, the stream and (for N > 1 GPUs) the
RCCL communicator.  All arithmetic of the training step happens in
lib/libnsfnet_pinn.so through the C ABI of include/nsfnet_pinn.h.

The step implemented is the reference's solve_Adam loop body
(NSFnet/pinn_solver.py:250-254, ev-NSFnet/pinn_solver.py:456-472):
loss (BC MSE + PDE residual MSE [+ supervised MSE]) -> d loss/d theta -> Adam.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib

NLOSS = 8
FLD = dict(u=0, v=1, u_x=2, u_y=3, v_x=4, v_y=5, eq1=6, eq2=7, eq3=8, eq4=9, p=10)
FLD_COUNT = 11

# slots of the per-step sums vector (all-reduced together with the gradients)
# (three blocks of NLOSS, each written whole by one loss-sum launch: no per-step zeroing or copies)
S_EQ = 0        # 0..3   sum w*eq_k^2
S_BC = 8        # 8,9    sum (u-u_b)^2, sum (v-v_b)^2
S_SUP = 16      # 16..18 sum sq err u,v,p ; 19 = number of finite p targets
NSUMS = 24


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def layer_shapes(n_out, n_hidden, hidden):
    widths = [2] + [hidden] * n_hidden + [n_out]
    return [(widths[i + 1], widths[i]) for i in range(len(widths) - 1)]


PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16": 2}


def resolve_precision(precision=None):
    """'fp32' | 'bf16x3' | 'bf16' or a (fwd, bwd, dw) triple of those; default from
    $NSFNET_PRECISION, else fp32 (the bit-exact fp32 MFMA path)."""
    import os
    if precision is None:
        precision = os.environ.get("NSFNET_PRECISION", "fp32")
    if isinstance(precision, str):
        precision = tuple(precision.split(",")) if "," in precision else (precision,) * 3
    if len(precision) != 3 or any(p not in PRECISIONS for p in precision):
        raise ValueError("precision must be one of %s (or a fwd,bwd,dw triple)" % sorted(PRECISIONS))
    return tuple(precision)


class DeviceNet:
    """One FCNet on the device: flat fp32 parameters in reference state_dict order
    (NSFnet/net.py:36-46) plus their MFMA-fragment-ordered copy."""

    def __init__(self, n_out, n_hidden, hidden, device, precision=None):
        self.lib = _lib.load()
        self.n_out, self.n_hidden, self.hidden, self.device = n_out, n_hidden, hidden, device
        self.precision = resolve_precision(precision)
        h = ctypes.c_void_p()
        _lib.check(self.lib.pinn_net_create(n_out, n_hidden, hidden, ctypes.byref(h)), "pinn_net_create")
        _lib.check(self.lib.pinn_net_set_precision(h, *[PRECISIONS[p] for p in self.precision]),
                   "pinn_net_set_precision")
        self.handle = h
        self.num_params = int(self.lib.pinn_net_num_params(h))
        self.params = torch.zeros(self.num_params, dtype=torch.float32, device=device)
        self.prep = torch.zeros(int(self.lib.pinn_net_prep_floats(h)), dtype=torch.float32, device=device)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.adam_t = 0                                                   # steps taken (host mirror)
        self.adam_t_dev = torch.zeros(2, dtype=torch.int64, device=device)   # [count, scratch] the kernel uses

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.pinn_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- state_dict interop (checkpoint format of the reference) ----
    def keys_and_shapes(self):
        out = []
        for i, (o, k) in enumerate(layer_shapes(self.n_out, self.n_hidden, self.hidden)):
            out.append(("layers.layer_%d.weight" % i, (o, k)))
            out.append(("layers.layer_%d.bias" % i, (o,)))
        return out

    def load_state_dict(self, sd):
        flat = []
        for key, shape in self.keys_and_shapes():
            t = sd[key]
            if tuple(t.shape) != tuple(shape):
                raise ValueError("state_dict[%s] has shape %s, expected %s" % (key, tuple(t.shape), shape))
            flat.append(t.detach().to(torch.float32).reshape(-1).cpu())
        self.set_flat(torch.cat(flat))

    def state_dict(self):
        sd, off = {}, 0
        flat = self.params.detach().cpu()
        for key, shape in self.keys_and_shapes():
            n = int(np.prod(shape))
            sd[key] = flat[off:off + n].reshape(shape).clone()
            off += n
        return sd

    def set_flat(self, flat):
        flat = torch.as_tensor(flat, dtype=torch.float32).reshape(-1)
        if flat.numel() != self.num_params:
            raise ValueError("expected %d parameters, got %d" % (self.num_params, flat.numel()))
        self.params.copy_(flat.to(self.device))
        self.prepare()

    def prepare(self):
        _lib.check(self.lib.pinn_net_prepare(self.handle, _ptr(self.params), _ptr(self.prep), _stream()),
                   "pinn_net_prepare")

    def reset_adam(self):
        self.m.zero_(); self.v.zero_(); self.adam_t = 0
        self.adam_t_dev.zero_()

    def adam_step(self, grads, lr, betas=(0.9, 0.999), eps=1e-8):
        """One Adam update + weight re-layout.  The step count lives on the device so the call is
        identical every step (hipGraph-capturable)."""
        self.adam_t += 1
        _lib.check(self.lib.pinn_adam_step_dev(_ptr(self.params), _ptr(grads), _ptr(self.m), _ptr(self.v),
                                               self.num_params, lr, betas[0], betas[1], eps,
                                               _ptr(self.adam_t_dev), _stream()), "pinn_adam_step_dev")
        self.prepare()


class PointPlan:
    """A fixed set of points evaluated by one DeviceNet (residual or value mode)."""

    def __init__(self, net, x, y, streams, with_backward=True, ws=None):
        self.lib, self.net, self.streams = net.lib, net, streams
        dev = net.device
        self.x = torch.as_tensor(np.asarray(x, dtype=np.float32).reshape(-1)).to(dev).contiguous()
        self.y = torch.as_tensor(np.asarray(y, dtype=np.float32).reshape(-1)).to(dev).contiguous()
        self.n = self.x.numel()
        if self.y.numel() != self.n or self.n == 0:
            raise ValueError("x and y must be non-empty and of equal length")
        h = ctypes.c_void_p()
        _lib.check(self.lib.pinn_plan_create(net.handle, self.n, streams, ctypes.byref(h)), "pinn_plan_create")
        self.handle = h
        self.npad = int(self.lib.pinn_plan_padded_points(h))
        self.with_backward = with_backward
        nbytes = int(self.lib.pinn_plan_workspace_bytes(h, 1 if with_backward else 0))
        if ws is not None and ws.numel() < nbytes:
            raise ValueError("shared workspace too small: %d < %d bytes" % (ws.numel(), nbytes))
        self.ws = ws if ws is not None else torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        self.sums = torch.zeros(NLOSS, dtype=torch.float32, device=dev)

    def kernel_names(self):
        """(forward, reverse sweep, weight-gradient) kernel family names this plan launches (for profiling)."""
        return tuple((self.lib.pinn_plan_kernel(self.handle, k) or b"").decode() for k in (0, 1, 2))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.pinn_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class ResidualPlan(PointPlan):
    def __init__(self, net, x, y, weights=None, with_backward=True, ws=None):
        super().__init__(net, x, y, 4, with_backward, ws)
        dev = net.device
        self.fields = torch.zeros(FLD_COUNT, self.npad, dtype=torch.float32, device=dev)
        self.w = None if weights is None else torch.as_tensor(
            np.asarray(weights, dtype=np.float32).reshape(-1)).to(dev).contiguous()
        if self.w is not None and self.w.numel() != self.n:
            raise ValueError("weights must have one entry per collocation point")
        self.vis_t = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.vis_t_minus = None     # lagged alpha_evm*|e| state (ev flavour)
        self.ebar = None

    def forward(self, Re, e=None, vis_t0=0.0, alpha_evm=0.0, scale=1.0, save=True, sums_out=None):
        if save and not self.with_backward:
            raise RuntimeError("plan was created without backward workspace")
        _lib.check(self.lib.pinn_residual_forward(
            self.handle, _ptr(self.ws), _ptr(self.net.prep), _ptr(self.x), _ptr(self.y), _ptr(e), _ptr(self.w),
            _ptr(self.vis_t_minus), _ptr(self.vis_t), _ptr(self.fields), float(Re), float(vis_t0),
            float(alpha_evm), float(scale), 1 if save else 0,
            _ptr(self.sums if sums_out is None else sums_out), _stream()), "pinn_residual_forward")

    def backward(self, Re, coef_eq, e=None, scale=1.0, want_ebar=False, phases=3):
        if want_ebar and self.ebar is None:
            self.ebar = torch.zeros(((self.n + 127) // 128) * 128, dtype=torch.float32, device=self.net.device)
        coef = (ctypes.c_float * 4)(*[float(c) for c in coef_eq])
        _lib.check(self.lib.pinn_residual_backward_phases(
            self.handle, _ptr(self.ws), _ptr(self.net.prep), _ptr(self.x), _ptr(self.y), _ptr(e), _ptr(self.w),
            _ptr(self.vis_t), _ptr(self.fields), coef, float(Re), float(scale),
            _ptr(self.ebar if want_ebar else None), int(phases), _stream()), "pinn_residual_backward")

    def field(self, name):
        return self.fields[FLD[name], :self.n]


class ChunkedResidual:
    """Collocation set processed in passes of `chunk_points` points that SHARE one activation workspace
    (forward -> reverse sweep -> gradient assembly per pass, gradients and loss sums accumulated), for
    point sets whose saved activations would not fit in HBM at once - the mini-batching of the
    reference's roadmap (ev-NSFnet/README.md:118; its `batchsize` argument is dead).  Full-batch
    semantics are unchanged: every pass uses the global normalisation.  Presents the attributes of a
    ResidualPlan (field(), vis_t, vis_t_minus, ebar, n)."""

    ALIGN = 128      # chunk boundaries on a multiple of every tile size (16 / 32 / 128 points)

    def __init__(self, net, x, y, weights=None, chunk_points=1 << 20):
        x = np.asarray(x, dtype=np.float32).reshape(-1)
        y = np.asarray(y, dtype=np.float32).reshape(-1)
        w = None if weights is None else np.asarray(weights, dtype=np.float32).reshape(-1)
        if w is not None and w.size != x.size:
            raise ValueError("weights must have one entry per collocation point")
        self.net, self.n = net, x.size
        chunk = max(self.ALIGN, int(chunk_points) // self.ALIGN * self.ALIGN)
        self.bounds = [(a, min(a + chunk, self.n)) for a in range(0, self.n, chunk)]
        self.chunks, ws = [], None
        for a, b in self.bounds:       # the first pass is the largest: it sizes the shared workspace
            c = ResidualPlan(net, x[a:b], y[a:b], None if w is None else w[a:b], ws=ws)
            ws = c.ws
            self.chunks.append(c)
        self.npad = sum(c.npad for c in self.chunks)
        self.w = None if w is None else torch.cat([c.w for c in self.chunks])
        self.tmp_sums = torch.zeros(NLOSS, dtype=torch.float32, device=net.device)

    def field(self, name):
        return torch.cat([c.field(name) for c in self.chunks])

    @property
    def vis_t(self):
        return torch.cat([c.vis_t for c in self.chunks])

    @property
    def vis_t_minus(self):
        if self.chunks[0].vis_t_minus is None:
            return None
        return torch.cat([c.vis_t_minus for c in self.chunks])

    @vis_t_minus.setter
    def vis_t_minus(self, t):
        for (a, b), c in zip(self.bounds, self.chunks):
            c.vis_t_minus = None if t is None else t.reshape(-1)[a:b].contiguous()

    @property
    def ebar(self):
        """d loss / d e for all points, padded like a ResidualPlan's (seed of the entropy-net backward)."""
        out = torch.zeros(((self.n + 127) // 128) * 128, dtype=torch.float32, device=self.net.device)
        for (a, b), c in zip(self.bounds, self.chunks):
            out[a:b] = c.ebar[:b - a]
        return out


class ValuePlan(PointPlan):
    def __init__(self, net, x, y, targets=None, with_backward=True):
        super().__init__(net, x, y, 1, with_backward)
        dev = net.device
        self.pred = torch.zeros(net.n_out, self.n, dtype=torch.float32, device=dev)
        self.targets = [None, None, None]
        if targets is not None:
            for c, t in enumerate(targets):
                if t is not None:
                    tt = torch.as_tensor(np.asarray(t, dtype=np.float32).reshape(-1)).to(dev).contiguous()
                    if tt.numel() != self.n:
                        raise ValueError("target %d must have one entry per point" % c)
                    self.targets[c] = tt

    def forward(self, coef=(0.0, 0.0, 0.0), save=False, use_targets=True, sums_out=None):
        if save and not self.with_backward:
            raise RuntimeError("plan was created without backward workspace")
        n_out = self.net.n_out
        pred = (ctypes.c_void_p * 3)(*[self.pred[c].data_ptr() if c < n_out else 0 for c in range(3)])
        tgt = (ctypes.c_void_p * 3)(*[(self.targets[c].data_ptr() if (use_targets and self.targets[c] is not None) else 0)
                                      for c in range(3)])
        cf = (ctypes.c_float * 3)(*[float(c) for c in coef])
        _lib.check(self.lib.pinn_value_forward(
            self.handle, _ptr(self.ws), _ptr(self.net.prep), _ptr(self.x), _ptr(self.y), pred, tgt, cf,
            1 if save else 0, _ptr(self.sums if sums_out is None else sums_out), _stream()), "pinn_value_forward")

    def backward(self, out_adj=None):
        _lib.check(self.lib.pinn_value_backward(
            self.handle, _ptr(self.ws), _ptr(self.net.prep), _ptr(self.x), _ptr(self.y), _ptr(out_adj), _stream()),
            "pinn_value_backward")


def grad_reduce(net, plans, grads_out, accumulate=False):
    lib = net.lib
    n = len(plans)
    ph = (ctypes.c_void_p * n)(*[p.handle.value for p in plans])
    wh = (ctypes.c_void_p * n)(*[p.ws.data_ptr() for p in plans])
    _lib.check(lib.pinn_grad_reduce(net.handle, n, ph, wh, _ptr(grads_out), 1 if accumulate else 0, _stream()),
               "pinn_grad_reduce")


class PinnEngine:
    """The per-step hot path for one rank.

    flavour 'nsfnet': loss = alpha_b*loss_b + alpha_e*(m(eq1)+m(eq2)+m(eq3)), nu = 1/Re
                      (NSFnet/pinn_solver.py:197-226)
    flavour 'ev':     adds the entropy net e, lagged artificial viscosity, eq4 with weight
                      0.1, SDF weights, optional supervised loss
                      (ev-NSFnet/pinn_solver.py:290-342, 372-428)
    Multi-GPU: every rank holds a shard of the points; ONE all-reduce (RCCL) of
    [grad | grad_e | sums] per step, all normalisations use GLOBAL counts.
    """

    def __init__(self, device, n_hidden, hidden, Re, alpha_b=1.0, alpha_e=1.0, flavour="nsfnet",
                 n_hidden_e=None, hidden_e=None, alpha_evm=0.0, alpha_s=0.0, coord_scale=1.0,
                 vis_t0_factor=20.0, process_group=None, world_size=1, net=None, net_e=None, precision=None):
        self.device = torch.device(device)
        self.flavour = flavour
        self.Re = float(Re)
        self.alpha_b, self.alpha_e, self.alpha_s = float(alpha_b), float(alpha_e), float(alpha_s)
        self.alpha_evm = float(alpha_evm)
        self.scale = float(coord_scale)
        self.vis_t0 = vis_t0_factor / self.Re
        self.net = net if net is not None else DeviceNet(3, n_hidden, hidden, self.device, precision)
        if flavour == "ev":
            self.net_e = net_e if net_e is not None else DeviceNet(1, n_hidden_e, hidden_e, self.device, precision)
        else:
            self.net_e = None
        self.e_trainable = False
        self.pg, self.world_size = process_group, int(world_size)
        P = self.net.num_params + (self.net_e.num_params if self.net_e else 0)
        self.flat = torch.zeros(P + NSUMS, dtype=torch.float32, device=self.device)
        self.P, self.P1 = self.net.num_params, (self.net_e.num_params if self.net_e else 0)
        self.plan_f = self.plan_b = self.plan_s = self.plan_e = None
        self._graphs = {}
        self._side = None
        import os
        self._overlap = os.environ.get("NSFNET_OVERLAP_BC", "1") not in ("0", "", "false")
        self.n_f_global = self.n_b_global = self.n_s_global = 0
        self._n_p_local, self._n_p_valid, self._sup_stale = 0, None, False
        self.eq4_weight = 0.1
        # 'MSE' (every script of the reference) or 'L2': 2-norms of the residual / boundary-misfit vectors
        # (NSFnet/pinn_solver.py:202-204, 214-217; plain NSFnet, one GPU).  The same kernels run: only the adjoint
        # coefficients change, from 2 alpha / N to alpha / ||r_k||, and the norms have to be known first - one
        # host read of the forward sums per evaluation (no hipGraph in this mode).
        self.loss_mode = "MSE"

    # ---- views into the exchange buffer ----
    @property
    def grads(self):
        return self.flat[:self.P]

    @property
    def grads_e(self):
        return self.flat[self.P:self.P + self.P1]

    @property
    def sums(self):
        return self.flat[self.P + self.P1:]

    # ---- data ----
    def set_collocation(self, x, y, weights=None, n_global=None, chunk_points=None):
        """chunk_points (or $NSFNET_CHUNK_POINTS): process the set in passes of that many points sharing
        one activation workspace (ChunkedResidual); default: one pass, everything resident."""
        import os
        self._graphs.clear()      # captured steps hold the old plan's pointers
        if chunk_points is None and os.environ.get("NSFNET_CHUNK_POINTS"):
            chunk_points = int(os.environ["NSFNET_CHUNK_POINTS"])
        n = int(np.asarray(x).size)
        if chunk_points and n > int(chunk_points):
            self.plan_f = ChunkedResidual(self.net, x, y, weights, chunk_points)
        else:
            self.plan_f = ResidualPlan(self.net, x, y, weights)
        self.n_f_global = int(n_global if n_global is not None else self.plan_f.n)
        if self.net_e is not None:
            self.plan_e = ValuePlan(self.net_e, x, y)
            self.init_vis_t()

    def set_boundary(self, x, y, u, v, n_global=None):
        self._graphs.clear()      # captured steps hold the old plan's pointers
        self.plan_b = ValuePlan(self.net, x, y, targets=[u, v, None])
        self.n_b_global = int(n_global if n_global is not None else self.plan_b.n)

    def set_supervised(self, x, y, u, v, p=None, n_global=None):
        """Supervised samples of THIS rank (ev-NSFnet/pinn_solver.py:202-251).  A rank's share may be
        empty (np.array_split with fewer samples than ranks, :219-221; the reference then skips the
        branch on that rank, :400): it contributes zero sums and no gradient but still takes part in
        the step's all-reduce and divides by the global counts."""
        self._graphs.clear()      # captured steps hold the old plan's pointers
        self._n_p_valid = None
        self.sums[S_SUP:S_SUP + NLOSS].zero_()
        self._sup_stale = False
        if x is None:
            self.plan_s, self.n_s_global, self._n_p_local = None, 0, 0
            return
        n_local = int(np.asarray(x).size)
        self.n_s_global = int(n_global if n_global is not None else n_local)
        if n_local == 0:
            self.plan_s, self._n_p_local = None, 0
            return
        self.plan_s = ValuePlan(self.net, x, y, targets=[u, v, p])
        self._n_p_local = 0 if p is None else int(np.isfinite(np.asarray(p, dtype=np.float64)).sum())

    def init_vis_t(self):
        """vis_t_minus = alpha_evm*|e(x_f)|   (ev-NSFnet/pinn_solver.py:138-140)"""
        self.plan_e.forward(save=False)
        fresh = (self.alpha_evm * self.plan_e.pred[0].abs()).contiguous()
        cur = self.plan_f.vis_t_minus
        if cur is not None and not isinstance(self.plan_f, ChunkedResidual) and cur.shape == fresh.shape:
            cur.copy_(fresh)          # in place: a captured hipGraph keeps reading / writing this allocation
        else:
            self._graphs.clear()
            self.plan_f.vis_t_minus = fresh

    # ---- one loss + gradient evaluation ----
    def loss_and_grad(self):
        f, b = self.plan_f, self.plan_b
        sums = self.sums
        sup_on = self.n_s_global > 0 and self.alpha_s != 0.0
        s = self.plan_s if sup_on else None             # None also on a rank whose supervised share is empty
        n_p = self._n_p_valid_global() if sup_on else 0
        if s is None and self._sup_stale:
            # the supervised block is all-reduced in place with everything else: what an earlier step left
            # there (this rank's own sums, or - on a rank with an empty share - the global sums) would be
            # added again, and multiplied by world_size, every step
            sums[S_SUP:S_SUP + NLOSS].zero_()
            self._sup_stale = False
        if sup_on:
            self._sup_stale = True
        # The value-mode chains (boundary / supervised points: a few thousand points, latency-bound
        # kernels) are independent of the collocation chain until the gradient assembly: they run on a
        # second HIP stream beside it.
        main = side = None
        if self.device.type == "cuda" and self._overlap:
            main = torch.cuda.current_stream(self.device)
            side = self._side_stream(main)
            side.wait_stream(main)
            torch.cuda.set_stream(side)
        l2 = self.loss_mode == "L2"
        if l2 and (self.net_e is not None or self.world_size > 1 or sup_on or isinstance(f, ChunkedResidual)):
            raise NotImplementedError("loss_mode 'L2' exists for the plain NSFnet flavour on one GPU (NSFnet/pinn_solver.py:202-217)")
        try:
            cb = 2.0 * self.alpha_b / self.n_b_global
            if l2:      # norms first (2052 boundary points: a forward-only pass), then the adjoints alpha_b (u - u_b) / ||u - u_b||
                b.forward(coef=(0.0, 0.0, 0.0), save=False, sums_out=sums[S_BC:S_BC + NLOSS])
                nb = torch.sqrt(sums[S_BC:S_BC + 2]).cpu().numpy().astype(np.float64)
                b.forward(coef=(self.alpha_b / max(nb[0], 1e-30), self.alpha_b / max(nb[1], 1e-30), 0.0), save=True,
                          sums_out=sums[S_BC:S_BC + NLOSS])
            else:
                b.forward(coef=(cb, cb, 0.0), save=True, sums_out=sums[S_BC:S_BC + NLOSS])
            b.backward()
            if s is not None:
                # per-output means: u,v over all supervised points, p over its finite targets (ev:399-411)
                cs = 2.0 * self.alpha_s / self.n_s_global
                s.forward(coef=(cs, cs, (2.0 * self.alpha_s / n_p) if n_p > 0 else 0.0), save=True,
                          sums_out=sums[S_SUP:S_SUP + NLOSS])
                s.backward()
        finally:
            if side is not None:
                torch.cuda.set_stream(main)
        e = None
        if self.net_e is not None:
            self.plan_e.forward(save=self.e_trainable)
            e = self.plan_e.pred[0]
        c = 2.0 * self.alpha_e / self.n_f_global
        coef_eq = (c, c, c, c * self.eq4_weight if self.net_e is not None else 0.0)
        value_plans = [b] if s is None else [b, s]
        if isinstance(f, ChunkedResidual):
            # one pass per chunk through the shared workspace; gradients / sums accumulate
            sums[S_EQ:S_EQ + NLOSS].zero_()
            for k, ((lo, hi_), ck) in enumerate(zip(f.bounds, f.chunks)):
                ek = None if e is None else e[lo:hi_]
                ck.forward(self.Re, e=ek, vis_t0=self.vis_t0, alpha_evm=self.alpha_evm, scale=self.scale, save=True,
                           sums_out=f.tmp_sums)
                sums[S_EQ:S_EQ + NLOSS] += f.tmp_sums
                ck.backward(self.Re, coef_eq, e=ek, scale=self.scale, want_ebar=self.e_trainable)
                grad_reduce(self.net, [ck], self.grads, accumulate=k > 0)
            if side is not None:
                main.wait_stream(side)
            grad_reduce(self.net, value_plans, self.grads, accumulate=True)
        else:
            f.forward(self.Re, e=e, vis_t0=self.vis_t0, alpha_evm=self.alpha_evm, scale=self.scale, save=True,
                      sums_out=sums[S_EQ:S_EQ + NLOSS])
            if l2:      # d ||eq_k|| / d theta = sum eq_k d eq_k / ||eq_k||: the reverse sweep's seeds with alpha_e / ||eq_k||
                ne = torch.sqrt(sums[S_EQ:S_EQ + 3]).cpu().numpy().astype(np.float64)
                coef_eq = tuple(self.alpha_e / max(v, 1e-30) for v in ne) + (0.0,)
            f.backward(self.Re, coef_eq, e=e, scale=self.scale, want_ebar=self.e_trainable)
            if side is not None:
                main.wait_stream(side)
            grad_reduce(self.net, [f] + value_plans, self.grads)
        if self.net_e is not None:
            if self.e_trainable:
                self.plan_e.backward(out_adj=f.ebar)
                grad_reduce(self.net_e, [self.plan_e], self.grads_e)
            else:
                self.grads_e.zero_()
        if self.world_size > 1:
            torch.distributed.all_reduce(self.flat, group=self.pg)

    def _side_stream(self, main):
        """The stream the value-mode chains run on.  Inside a graph capture it must be a stream that is
        not already capturing something else; one persistent stream per engine serves both cases."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _n_p_valid_global(self):
        if getattr(self, "_n_p_valid", None) is None:
            n = int(self._n_p_local)
            if self.world_size > 1:
                tt = torch.tensor([n], dtype=torch.int64, device=self.device)
                torch.distributed.all_reduce(tt, group=self.pg)
                n = int(tt.item())
            self._n_p_valid = n
        return self._n_p_valid

    def loss_terms(self):
        """Device tensors (no sync): dict of loss_eq1..4, loss_e, loss_b, loss_s, loss."""
        s = self.sums
        if self.loss_mode == "L2":      # NSFnet/pinn_solver.py:202-204, 214-217
            eq = torch.sqrt(s[S_EQ:S_EQ + 4])
            loss_e = eq[0] + eq[1] + eq[2]
            loss_b = torch.sqrt(s[S_BC]) + torch.sqrt(s[S_BC + 1])
            return dict(loss_eq1=eq[0], loss_eq2=eq[1], loss_eq3=eq[2], loss_eq4=eq[3], loss_e=loss_e, loss_b=loss_b,
                        loss_s=torch.zeros((), device=self.device), loss=self.alpha_b * loss_b + self.alpha_e * loss_e)
        eq = s[S_EQ:S_EQ + 4] / self.n_f_global
        loss_e = eq[0] + eq[1] + eq[2] + (self.eq4_weight * eq[3] if self.net_e is not None else 0.0)
        loss_b = (s[S_BC] + s[S_BC + 1]) / self.n_b_global
        out = dict(loss_eq1=eq[0], loss_eq2=eq[1], loss_eq3=eq[2], loss_eq4=eq[3], loss_e=loss_e, loss_b=loss_b)
        loss_s = torch.zeros((), device=self.device)
        if self.n_s_global > 0 and self.alpha_s != 0.0:
            n_p = self._n_p_valid_global()
            loss_s = (s[S_SUP] + s[S_SUP + 1]) / self.n_s_global + (s[S_SUP + 2] / n_p if n_p > 0 else 0.0)
        out["loss_s"] = loss_s
        out["loss"] = self.alpha_b * loss_b + self.alpha_e * loss_e + self.alpha_s * loss_s
        return out

    def adam_step(self, lr):
        self.net.adam_step(self.grads, lr)
        if self.net_e is not None and self.e_trainable:
            self.net_e.adam_step(self.grads_e, lr)

    def step(self, lr):
        """loss + gradient + (all-reduce) + Adam.  With NSFNET_GRAPH=1 the launch sequence is captured
        once per (lr, schedule state) in a hipGraph and replayed.  Opt-in: measured on MI355X the eager
        launch sequence (~14 launches, all asynchronous) already keeps the GPU busy down to the 4x50 /
        10 k-point step (0.12 ms), and replay is 0-6 % slower; it pays only when the host is contended."""
        if not self._graphs_enabled() or self.loss_mode != "MSE":
            self.loss_and_grad()
            self.adam_step(lr)
            return
        key = (float(lr), self.e_trainable, self.alpha_evm, self.alpha_b, self.alpha_e, self.alpha_s, self.scale,
               self.n_f_global, self.n_b_global, self.n_s_global, self.Re, self.vis_t0, self.eq4_weight)
        g = self._graphs.get(key)
        if g is None:
            # first use of this configuration: run it eagerly once (lazy host-side setup such as the
            # supervised-target census happens here), then capture
            self.loss_and_grad()
            self.adam_step(lr)
            if len(self._graphs) >= 8:
                self._graphs.clear()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    self.loss_and_grad()
                    self.adam_step(lr)
            torch.cuda.current_stream(self.device).wait_stream(side)
            # the capture itself does not execute; account for the host mirror it advanced
            self.net.adam_t -= 1
            if self.net_e is not None and self.e_trainable:
                self.net_e.adam_t -= 1
            self._graphs[key] = graph
            return
        g.replay()
        self.net.adam_t += 1
        if self.net_e is not None and self.e_trainable:
            self.net_e.adam_t += 1

    def _graphs_enabled(self):
        import os
        flag = os.environ.get("NSFNET_GRAPH")
        return flag is not None and flag not in ("0", "", "false", "False") and self.device.type == "cuda"

    # ---- inference (evaluate / test / predict) ----
    def predict(self, x, y, with_e=False):
        plan = ValuePlan(self.net, x, y, with_backward=False)
        plan.forward(save=False)
        out = [plan.pred[0], plan.pred[1], plan.pred[2]]
        if with_e and self.net_e is not None:
            pe = ValuePlan(self.net_e, x, y, with_backward=False)
            pe.forward(save=False)
            out.append(pe.pred[0])
        return out
