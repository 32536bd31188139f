"""Errors of the default bf16x3 path (role-split sweeps, 24-bit spill) against the fp64 oracle at 6x256:
    python scripts/precision_report.py          (needs the GPU; PINN_SCHED=0 gives the 8-wave kernels with the fp32 spill)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import autograd_ref as ar, fwdmode_ref as fr
from nsfnet_amd import engine as eng
L, H, N = 6, 256, 4096
dev = torch.device("cuda:0")
flat = ar.flat_params(ar.seeded_net(3, L, H, seed=46)).numpy().copy()
rng = np.random.RandomState(7)
x = rng.rand(N).astype(np.float32); y = rng.rand(N).astype(np.float32)
P = fr.unflatten(flat.astype(np.float64), 2, 3, L, H)
r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), 2000.0, alpha_e=1.0)
for prec in ("fp32", "bf16x3"):
    E = eng.PinnEngine(dev, L, H, 2000.0, alpha_b=0.0, alpha_e=1.0, precision=prec)
    E.net.set_flat(torch.tensor(flat)); E.set_collocation(x, y)
    xb = np.array([0.5], np.float32); E.set_boundary(xb, xb, xb * 0, xb * 0)
    E.loss_and_grad(); torch.cuda.synchronize()
    f = E.plan_f.fields[:, :N].cpu().numpy().astype(np.float64)
    res = max(np.abs(f[6 + k] - r["eqs"][k]).max() / np.abs(r["eqs"][k]).max() for k in range(3))
    sums = np.abs(E.sums.cpu().numpy()[:3] - r["sums"]) / np.abs(r["sums"])
    g = E.grads.cpu().numpy().astype(np.float64)
    print("%-7s kernels %s: residuals %.2e of max, loss sums %.2e, gradient rel-L2 %.2e" % (
        prec, E.plan_f.kernel_names()[:2], res, sums.max(), np.linalg.norm(g - r["grad"]) / np.linalg.norm(r["grad"])))
