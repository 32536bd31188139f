"""Diagnostic: per-slot s_memtime stamps of the pipelined forward (library built with -DPINN_STAMP).
    NSFNET_PINN_LIB=experiments/abl/lib_stamp.so python scripts/stamp_fwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nsfnet_amd import engine as eng
dev = torch.device("cuda:0")
L, H, Re = 6, 256, 2000.0
x, y = bench.grid_block(600, 600, 0, 1)
e = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0, precision="bf16x3")
e.net.set_flat(bench.seeded_flat(L, H))
e.set_collocation(x, y)
f = e.plan_f
buf = torch.zeros(max(f.n, 4096), dtype=torch.float32, device=dev)
for _ in range(3):
    f.forward(Re, e=buf, save=True)
torch.cuda.synchronize()
st = buf[:4 * 64 * 2].view(torch.int64).cpu().numpy().reshape(4, 64)
names = ["start", "mid-", "mid+", "end"]
for w in range(4):
    t = st[w]
    t0 = t[0]
    print("wave", w)
    for s in range(12):
        q = t[4 * s:4 * s + 4] - t0
        nxt = t[4 * s + 4] - t0 if 4 * s + 4 < 64 and t[4 * s + 4] > 0 else -1
        print("  slot %2d: start %8d  ss1 %6d  midbar %5d  ss2 %6d  | endbar+gap %6d" % (
            s, q[0], q[1] - q[0], q[2] - q[1], q[3] - q[2], (nxt - q[3]) if nxt >= 0 else -1))
