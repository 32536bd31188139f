"""A/B/C of the 8-wave (0), pipelined (1) and role-split (2) bf16x3 sweeps at the headline shape, interleaved rounds in ONE process
(cdna_hip_programming.md rule 24): per-kernel HIP-event times, min and median over rounds.
    python scripts/ab_pipe.py [--grid 600] [--rounds 5] [--prec bf16x3]"""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=600)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--prec", default="bf16x3")
    ap.add_argument("--layers", type=int, default=6)
    a = ap.parse_args()
    from nsfnet_amd import engine as eng
    dev = torch.device("cuda:0")
    L, H, Re = a.layers, 256, 2000.0
    x, y = bench.grid_block(a.grid, a.grid, 0, 1)
    xb, yb, ub, vb = bench.cavity_boundary()
    E = {}
    for pipe in ("0", "1", "2"):
        os.environ["PINN_SCHED"] = pipe
        e = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0, precision=a.prec)
        e.net.set_flat(bench.seeded_flat(L, H))
        e.set_collocation(x, y); e.set_boundary(xb, yb, ub, vb)
        for _ in range(3):
            e.step(1e-3)
        E[pipe] = e
    torch.cuda.synchronize()
    n = x.size
    c = 2.0 / n
    res = {k: dict(fwd=[], bwd=[], dw=[], step=[]) for k in E}
    for r in range(a.rounds):
        for k, e in E.items():
            f = e.plan_f
            res[k]["fwd"].append(bench.time_kernel(lambda: f.forward(Re, save=True), 5))
            res[k]["bwd"].append(bench.time_kernel(lambda: f.backward(Re, (c, c, c, 0.0), phases=1), 5))
            res[k]["dw"].append(bench.time_kernel(lambda: f.backward(Re, (c, c, c, 0.0), phases=2), 5))
            res[k]["step"].append(bench.time_kernel(lambda: e.step(1e-3), 10))
    for k in E:
        print("PINN_SCHED=%s  " % k + "  ".join("%s min %.3f med %.3f" % (n_, min(v), float(np.median(v))) for n_, v in res[k].items()),
              " loss %.6f" % float(E[k].loss_terms()["loss"]), flush=True)


if __name__ == "__main__":
    main()
