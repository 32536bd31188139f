"""ev-NSFnet flavour step timing (BASELINE config 4 shape on one GPU: 6x256 + 4x40 entropy net)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from nsfnet_amd import engine as eng
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda:0")
E = eng.PinnEngine(dev, 6, 256, 4000.0, alpha_b=10.0, alpha_e=1.0, flavour="ev", n_hidden_e=4, hidden_e=40,
                   alpha_evm=0.05, precision=prec)
E.net.set_flat(bench.seeded_flat(6, 256)); E.net_e.set_flat(bench.seeded_flat(4, 40, n_out=1, seed=4321))
x, y = bench.grid_block(grid, grid, 0, 1)
E.set_collocation(x, y)
xb, yb, ub, vb = bench.cavity_boundary()
E.set_boundary(xb, yb, ub, vb)
for trainable in (False, True):
    E.e_trainable = trainable
    for _ in range(3): E.step(1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): E.step(1e-3)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("ev %s e_trainable=%s: %.2f ms/step = %.3g pts/s (loss %.5f)" % (prec, trainable, dt * 1e3, grid * grid / dt, float(E.loss_terms()["loss"])), flush=True)
t = bench.time_kernel(lambda: E.plan_e.forward(save=False), 5)
print("entropy-net forward alone: %.3f ms" % t)
