"""Random-shape sweep of the HIP step against the fp64 oracle (GPU): depth 1-9, width 3-512, ragged point
counts, all precision modes.  Prints the worst loss / gradient error per mode; exits non-zero on a miss."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nsfnet_amd import engine as eng
from oracle import autograd_ref as ar, fwdmode_ref as fr

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = torch.device("cuda:0")
xb, yb, ub, vb = (a.reshape(-1)[::64].astype(np.float32) for a in ar.cavity_boundary())
worst = {}
fail = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    L = int(rng.randint(1, 10)); H = int(rng.choice([3, 17, 32, 50, 64, 80, 100, 128, 200, 256, 270, 300, 340, 360, 400, 440, 470, 512]))      # (257..448: the wide role-split sweeps)
    N = int(rng.randint(1, 700)); Re = float(rng.choice([100.0, 2000.0, 10000.0]))
    if L * max(H, 32) > 9 * 512 or (H > 256 and L > 8):
        continue
    flat = (ar.flat_params(ar.seeded_net(3, L, H, seed=int(rng.randint(1 << 30)))).numpy()).copy()
    x, y = rng.rand(N).astype(np.float32), rng.rand(N).astype(np.float32)
    P = fr.unflatten(flat.astype(np.float64), 2, 3, L, H)
    r = fr.pde_loss_and_grad(P, x.astype(np.float64), y.astype(np.float64), Re)
    b = fr.bc_loss_and_grad(P, xb.astype(np.float64), yb.astype(np.float64), ub, vb, alpha_b=10.0)
    ref_loss = 10.0 * sum(b["sums"]) / len(xb) + sum(r["sums"]) / N
    ref_g = r["grad"] + b["grad"]
    for prec, tol_l, tol_g in (("fp32", 2e-5, 1e-4), ("bf16x3", 1e-4, 2e-4)):
        E = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0, precision=prec)
        E.net.set_flat(torch.tensor(flat))
        E.set_collocation(x, y); E.set_boundary(xb, yb, ub, vb)
        E.loss_and_grad(); torch.cuda.synchronize()
        el = abs(float(E.loss_terms()["loss"]) - ref_loss) / ref_loss
        eg = np.linalg.norm(E.grads.cpu().numpy() - ref_g) / max(np.linalg.norm(ref_g), 1e-300)
        w = worst.setdefault(prec, [0.0, 0.0])
        w[0] = max(w[0], el); w[1] = max(w[1], eg)
        if not (el < tol_l and eg < tol_g):
            fail += 1
            print("MISS", prec, dict(L=L, H=H, N=N, Re=Re), "loss err %.2e grad err %.2e" % (el, eg), flush=True)
        del E
print("worst (loss rel, grad rel-L2):", {k: ["%.1e" % v for v in w] for k, w in worst.items()}, "misses:", fail)
sys.exit(1 if fail else 0)
