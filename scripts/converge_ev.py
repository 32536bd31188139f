"""Resumable ev-NSFnet convergence run against the DNS field (SURVEY 8d "converged field" gate).

The reference's production schedule (ev-NSFnet/configs/production.yaml: six Adam stages, alpha_evm
0.05 -> 0.002, lr 1e-3 -> 2e-6, 6x80 + 4x40 nets, 120 k LHS points, SDF weights) on ONE MI355X, in
slices that fit a 20-minute GPU job: every call runs stages [--first, --last] and writes the two
state_dicts + a JSON line per stage; the next call resumes from them (the reference re-creates Adam at
every stage start, so a stage boundary is an exact resume point for the optimiser; the lagged
viscosity state restarts from alpha*|e| as at the start of a run).

    python scripts/converge_ev.py --re 3000 --dns tests/golden/dns/cavity_Re3000_256_Uniform.mat \
        --out gpurun_out/conv_ev --first 1 --last 2 [--resume DIR] [--epochs-scale 0.5]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

STAGES = [(0.05, 500000, 1e-3), (0.03, 500000, 2e-4), (0.01, 500000, 4e-5),
          (0.005, 500000, 1e-5), (0.002, 500000, 2e-6), (0.002, 500000, 2e-6)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--re", type=int, default=3000)
    ap.add_argument("--dns", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--resume", default=None, help="directory holding net.pth / net_evm.pth of the previous slice")
    ap.add_argument("--first", type=int, default=1)
    ap.add_argument("--last", type=int, default=6)
    ap.add_argument("--epochs-scale", type=float, default=1.0)
    ap.add_argument("--nf", type=int, default=120000)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--hidden", type=int, default=80)
    a = ap.parse_args()
    from nsfnet_amd import ev_pinn_solver as es, cavity_data as cavity
    os.makedirs(a.out, exist_ok=True)
    dns = os.path.abspath(a.dns)
    res = None if a.resume is None else os.path.abspath(a.resume)
    os.chdir(a.out)
    np.random.seed(1234); torch.manual_seed(1234)       # same point set and (first slice) same init in every call
    P = es.PysicsInformedNeuralNetwork(
        Re=a.re, layers=a.layers, layers_1=4, hidden_size=a.hidden, hidden_size_1=40, N_f=a.nf, alpha_evm=0.05,
        bc_weight=10, eq_weight=1, supervised_data_weight=0.0,
        net_params=None if res is None else os.path.join(res, "net.pth"),
        net_params_1=None if res is None else os.path.join(res, "net_evm.pth"))
    P.log_interval = 20000
    from types import SimpleNamespace
    loader = cavity.EvDataLoader(path="./datasets/", N_f=a.nf, N_b=1000, sort_training_points=False,
                                 sdf_weighting=SimpleNamespace(enabled=True, min_weight=0.2, decay=5.0),
                                 coord_transform=False)
    P.set_boundary_data(X=loader.loading_boundary_data())
    xf, yf = loader.loading_training_data()
    P.set_coordinate_transform(loader.get_coord_scale())
    P.set_eq_training_data(X=(xf, yf), weights=loader.get_sdf_weights())
    P.clear_supervised_data(); P.set_supervised_loss_weight(0.0)
    P.save = lambda *args, **kw: None                   # no per-10 000-step checkpoints: one per stage below
    star = loader.loading_evaluate_data(dns)
    for k in range(a.first, a.last + 1):
        alpha, epochs, lr = STAGES[k - 1]
        n = max(1, int(epochs * a.epochs_scale))
        P.current_stage = "Stage %d" % k
        P.set_alpha_evm(alpha)
        t0 = time.time()
        P.train(num_epoch=n, lr=lr)
        torch.cuda.synchronize()
        dt = time.time() - t0
        eu, ev = P.evaluate(*star)[:2]
        rec = dict(stage=k, alpha_evm=alpha, lr=lr, steps=n, seconds=round(dt, 1), ms_per_step=round(1e3 * dt / n, 4),
                   loss=float(P.loss), loss_b=float(P.loss_b), loss_e=float(P.loss_e),
                   err_u=float(eu), err_v=float(ev), Re=a.re, net="%dx%d+4x40" % (a.layers, a.hidden), N_f=a.nf,
                   precision=os.environ.get("NSFNET_PRECISION", "fp32"))
        with open("stages.jsonl", "a") as fh:
            fh.write(json.dumps(rec) + "\n")
        print("STAGE", json.dumps(rec), flush=True)
        torch.save(P.net.state_dict(), "net.pth")
        torch.save(P.net_1.state_dict(), "net_evm.pth")


if __name__ == "__main__":
    main()
