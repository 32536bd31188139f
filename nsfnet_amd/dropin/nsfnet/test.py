#!/usr/bin/env python3
"""Evaluate saved NSFnet checkpoints on a DNS field (reference: NSFnet/test.py:23-76):
relative L2 errors + cavity_result_loop_<n>.mat for every checkpoint given."""
import argparse
import re

import cavity_data as cavity
import pinn_solver as psolver


def evaluate(args, net_params, loop):
    PINN = psolver.PysicsInformedNeuralNetwork(Re=args.re, layers=args.layers, hidden_size=args.hidden, N_f=args.nf,
                                               bc_weight=10, eq_weight=1, net_params=net_params)
    star = cavity.DataLoader(N_f=args.nf).loading_evaluate_data(args.data)
    PINN.evaluate(*star)
    PINN.test(*star, loop)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoints", nargs="+", help="model_cavity_loop_<n>.pth files")
    ap.add_argument("--data", required=True)
    ap.add_argument("--re", type=float, default=2000)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--hidden", type=int, default=120)
    ap.add_argument("--nf", type=int, default=40000)
    a = ap.parse_args()
    for ck in a.checkpoints:
        m = re.search(r"(\d+)\.pth", ck)
        evaluate(a, ck, int(m.group(1)) if m else 0)
