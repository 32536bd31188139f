"""Time the residual forward / reverse sweep / dW kernels of ONE library build (NSFNET_PINN_LIB) at the headline shape.
    NSFNET_PINN_LIB=experiments/abl/lib_noS.so python scripts/abl_time.py [--what fwd,bwd,dw] [--tag noS]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="600", help="G (= GxG) or AxB")
ap.add_argument("--what", default="fwd")
ap.add_argument("--tag", default="")
ap.add_argument("--prec", default="bf16x3")
ap.add_argument("--layers", type=int, default=6)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--zero", action="store_true", help="all parameters zero: every MFMA operand is zero (power / clock experiment)")
a = ap.parse_args()
from nsfnet_amd import engine as eng
dev = torch.device("cuda:0")
L, H, Re = a.layers, a.hidden, 2000.0
g = [int(v) for v in str(a.grid).lower().split("x")]
x, y = bench.grid_block(g[0], g[-1], 0, 1)
xb, yb, ub, vb = bench.cavity_boundary()
e = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0, precision=a.prec)
e.net.set_flat(bench.seeded_flat(L, H) * (0.0 if a.zero else 1.0))
e.set_collocation(x, y); e.set_boundary(xb, yb, ub, vb)
f = e.plan_f
c = 2.0 / x.size
fn = dict(fwd=lambda: f.forward(Re, save=True), bwd=lambda: f.backward(Re, (c, c, c, 0.0), phases=1),
          dw=lambda: f.backward(Re, (c, c, c, 0.0), phases=2), step=lambda: e.step(1e-3))
for _ in range(3):
    f.forward(Re, save=True); f.backward(Re, (c, c, c, 0.0))
torch.cuda.synchronize()
out = []
for k in a.what.split(","):
    t = [bench.time_kernel(fn[k], 8) for _ in range(4)]
    out.append("%s min %.3f med %.3f" % (k, min(t), float(np.median(t))))
print("%-12s %s" % (a.tag or os.environ.get("NSFNET_PINN_LIB", "product"), "  ".join(out)), flush=True)
