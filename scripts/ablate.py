"""Ablation timings of the residual forward (GPU): with / without the activation spill."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nsfnet_amd import engine as eng
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    return bench.time_kernel(fn, reps)

for prec in sys.argv[1:] or ["bf16x3"]:
    E = eng.PinnEngine(torch.device("cuda:0"), 6, 256, 2000.0, alpha_b=10.0, alpha_e=1.0, precision=prec)
    E.net.set_flat(bench.seeded_flat(6, 256))
    x, y = bench.grid_block(600, 600, 0, 1)
    E.set_collocation(x, y)
    f = E.plan_f
    c = 2.0 / f.n
    print(prec, "fwd save=1 %.3f ms | fwd save=0 %.3f ms | bwd %.3f | dw %.3f" % (
        t(lambda: f.forward(2000.0, save=True)), t(lambda: f.forward(2000.0, save=False)),
        t(lambda: f.backward(2000.0, (c, c, c, 0.0), phases=1)), t(lambda: f.backward(2000.0, (c, c, c, 0.0), phases=2))), flush=True)
    del E, f
    torch.cuda.empty_cache()
