"""Measure the error of each precision mode / kernel family against the fp64 oracle (GPU)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nsfnet_amd import engine as eng
from oracle import autograd_ref as ar, fwdmode_ref as fr

def run(L, H, N, prec, seed=7, Re=2000.0):
    flat = ar.flat_params(ar.seeded_net(3, L, H, seed=1234)).numpy().copy()
    rng = np.random.RandomState(seed)
    x = rng.rand(N).astype(np.float32); y = rng.rand(N).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[::8].astype(np.float32) for a in ar.cavity_boundary())
    E = eng.PinnEngine(torch.device("cuda:0"), L, H, Re, alpha_b=10.0, alpha_e=1.0, precision=prec)
    E.net.set_flat(torch.tensor(flat)); E.set_collocation(x, y); E.set_boundary(xb, yb, ub, vb)
    E.loss_and_grad(); torch.cuda.synchronize()
    P = fr.unflatten(flat.astype(np.float64), 2, 3, L, H)
    r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), Re)
    b = fr.bc_loss_and_grad(P, xb.astype(np.float64), yb.astype(np.float64), ub, vb, alpha_b=10.0)
    f = E.plan_f
    eqe = [np.abs(f.field("eq%d" % (k + 1)).cpu().numpy() - r["eqs"][k]).max() / np.abs(r["eqs"][k]).max() for k in range(3)]
    s = E.sums.cpu().numpy()
    se = [abs(s[k] - r["sums"][k]) / r["sums"][k] for k in range(3)]
    ref_loss = 10.0 * sum(b["sums"]) / len(xb) + sum(r["sums"]) / N
    le = abs(float(E.loss_terms()["loss"]) - ref_loss) / ref_loss
    g = E.grads.cpu().numpy().astype(np.float64); gr = r["grad"] + b["grad"]
    ge = np.linalg.norm(g - gr) / np.linalg.norm(gr)
    print("%dx%d N=%d %-22s eq maxrel %.1e %.1e %.1e | sums rel %.1e %.1e %.1e | loss rel %.1e | grad relL2 %.1e"
          % (L, H, N, ",".join(prec), *eqe, *se, le, ge), flush=True)

if __name__ == "__main__":
    shapes = [(4, 50, 1000), (6, 128, 1000), (6, 256, 1000)]
    modes = [("fp32",) * 3, ("bf16x3", "fp32", "fp32"), ("fp32", "bf16x3", "fp32"), ("fp32", "fp32", "bf16x3"),
             ("bf16x3",) * 3, ("bf16",) * 3]
    for (L, H, N) in shapes:
        for m in modes:
            run(L, H, N, m)
