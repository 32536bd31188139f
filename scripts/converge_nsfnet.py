#!/usr/bin/env python3
"""Plain-NSFnet convergence run (row g of the grading table): the reference's own schedule
(NSFnet/train.py:23-76 - Re = 2000, 4x120 net, 40 000 LHS points sorted by wall distance, lam_bcs = 10, five Adam
stages 200k/200k/200k/500k/500k at lr 1e-3 .. 2e-6) through the drop-in script on ONE MI355X, keeping only what
the topology test needs: the final state_dict, the relative L2 errors against DNS after every stage, the
residual losses and the vortex topology (scripts/flow_topology.py).

    python scripts/converge_nsfnet.py --out gpurun_out/conv_nsfnet [--epochs-scale 1.0] [--seed 1234]
"""
import argparse, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "nsfnet_amd", "dropin", "nsfnet"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--epochs-scale", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--precision", default="bf16x3")
    a = ap.parse_args()
    os.environ.setdefault("NSFNET_PRECISION", a.precision)
    out = os.path.abspath(a.out)
    os.makedirs(out, exist_ok=True)
    dns = os.path.join(ROOT, "tests", "golden", "dns", "cavity_Re2000_256.mat")
    import train as T, cavity_data as cavity, pinn_solver as psolver
    import flow_topology as ft
    np.random.seed(a.seed); torch.manual_seed(a.seed)
    work = tempfile.mkdtemp(prefix="nsfnet_conv_")
    os.chdir(work)
    P = psolver.PysicsInformedNeuralNetwork(Re=2000, layers=4, hidden_size=120, N_f=40000, bc_weight=10, eq_weight=1,
                                            num_ins=2, num_outs=3)
    P.save_every = 0              # no per-10k checkpoints: only the end state travels back
    P.log_every = 20000
    loader = cavity.DataLoader(path='./datasets/', N_f=40000, N_b=1000)
    P.set_boundary_data(X=loader.loading_boundary_data())
    P.set_eq_training_data(X=loader.loading_training_data())
    star = loader.loading_evaluate_data(dns)
    t0 = time.time()
    rec = []
    with open(os.path.join(out, "stages.jsonl"), "w") as fh:
        for k, (epochs, lr) in enumerate(T.STAGES, 1):
            P.set_stage(k)
            n = max(1, int(epochs * a.epochs_scale))
            P.train(num_epoch=n, lr=lr)
            eu, ev = P.evaluate(*star)
            r = dict(stage=k, steps=n, lr=lr, seconds=time.time() - t0, err_u=float(eu), err_v=float(ev),
                     loss=float(P.loss), loss_b=float(P.loss_b), loss_eq1=float(P.loss_eq1), loss_eq2=float(P.loss_eq2),
                     loss_eq3=float(P.loss_eq3))
            rec.append(r)
            fh.write(json.dumps(r) + "\n"); fh.flush()
            print(json.dumps(r), flush=True)
    torch.save(P.net.state_dict(), os.path.join(out, "nsfnet_re2000_4x120_net.pth"))
    X, Y, U, V = ft.load_dns(dns)
    u, v, p = P.engine.predict(X.reshape(-1), Y.reshape(-1))
    t = ft.topology(X, Y, u.cpu().numpy().reshape(X.shape).astype(np.float64), v.cpu().numpy().reshape(X.shape).astype(np.float64))
    json.dump(dict(stages=rec, topology=t, dns=ft.topology(X, Y, U, V), seed=a.seed, precision=a.precision),
              open(os.path.join(out, "summary.json"), "w"), indent=1)
    print(ft.describe("NSFnet end state", t))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
