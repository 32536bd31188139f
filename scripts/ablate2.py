"""Timing-only component ablation of the bf16x3 forward kernel (PINN_DBG bits; outputs are wrong by design)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from nsfnet_amd import engine as eng
E = eng.PinnEngine(torch.device("cuda:0"), 6, 256, 2000.0, alpha_b=10.0, alpha_e=1.0, precision="bf16x3")
E.net.set_flat(bench.seeded_flat(6, 256))
x, y = bench.grid_block(600, 600, 0, 1)
E.set_collocation(x, y)
f = E.plan_f
for dbg in (0, 1, 2, 4, 8, 3, 6, 7, 14):
    os.environ["PINN_DBG"] = str(dbg)
    f.forward(2000.0, save=True); torch.cuda.synchronize()
    t = bench.time_kernel(lambda: f.forward(2000.0, save=True), 5)
    print("dbg=%d (skip:%s%s%s) fwd %.3f ms" % (dbg, " mfma" if dbg & 1 else "", " stores" if dbg & 2 else "", " ldsw" if dbg & 4 else "", t) + (" [no W streaming]" if dbg & 8 else ""), flush=True)
