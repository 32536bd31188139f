"""Diagnostic: s_memtime stamps of the role-split forward sweep (library built with -DPINN_STAMP):
    python scripts/abl_build.py stamp="-DPINN_STAMP"
    NSFNET_PINN_LIB=experiments/abl/lib_stamp.so python scripts/stamp_fwd_split.py
Prints, for wave 0 of each group of workgroup 0 during its third pair of tiles, the duration of every quarter of every
phase: E quarters as (top -> quad 0 maths done -> quad 1 maths done -> stores issued) + barrier wait, G quarters as
MFMA run + barrier wait.  Cycles are s_memtime ticks (100 MHz constant clock on gfx950 -> printed in ns)."""
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
c = 2.0 / f.n
buf = torch.zeros(max(f.n, 8192), dtype=torch.float32, device=dev)
for _ in range(3):
    f.forward(Re, e=buf, save=True)
torch.cuda.synchronize()
st = buf[:4096].view(torch.int64).cpu().numpy().reshape(2, 1024)
for g in range(2):
    t = st[g]
    t0 = t[0]
    print("group", g, "(ticks since first stamp; tick = s_memtime unit)")
    i = 0
    for ph in range(2 * L - 1):
        if ph % 2 == 0:
            print("  E_%d" % (ph // 2))
            for q in range(4):
                a = t[i:i + 4]; nxt = t[i + 4]
                print("    q%d: start %7d  quad0 %5d  quad1 %5d  tail %5d | barrier %5d" % (
                    q, a[0] - t0, a[1] - a[0], a[2] - a[1], a[3] - a[2], nxt - a[3] if nxt > 0 else -1))
                i += 4
        else:
            print("  M_%d" % (ph // 2 + 1))
            for q in range(4):
                a = t[i:i + 2]; nxt = t[i + 2]
                print("    q%d: start %7d  mfma %5d | barrier %5d" % (q, a[0] - t0, a[1] - a[0], nxt - a[1] if nxt > 0 else -1))
                i += 2
