#!/usr/bin/env python3
"""Plain NSFnet training run: the reference's schedule (NSFnet/train.py:23-76) - Re=2000,
4x120 net, 40k LHS points sorted by wall distance, lam_bcs=10, five Adam stages - on the
MI355X engine.  --epochs-scale shrinks every stage (smoke runs); --hidden/--layers/--nf/--re
override the literals the reference hard-codes."""
import argparse
import os

import cavity_data as cavity
import pinn_solver as psolver

STAGES = [(200000, 1e-3), (200000, 2e-4), (200000, 5e-5), (500000, 1e-5), (500000, 2e-6)]


def train(args, net_params=None):
    PINN = psolver.PysicsInformedNeuralNetwork(
        Re=args.re, layers=args.layers, hidden_size=args.hidden, N_f=args.nf, bc_weight=10, eq_weight=1,
        num_ins=2, num_outs=3, net_params=net_params, checkpoint_path='./checkpoint/')
    loader = cavity.DataLoader(path='./datasets/', N_f=args.nf, N_b=1000)
    PINN.set_boundary_data(X=loader.loading_boundary_data())
    PINN.set_eq_training_data(X=loader.loading_training_data())
    ref = args.data or './data/cavity_Re%d_256.mat' % int(args.re)
    star = loader.loading_evaluate_data(ref) if os.path.exists(ref) else None
    for k, (epochs, lr) in enumerate(STAGES, 1):
        PINN.set_stage(k)
        PINN.train(num_epoch=max(1, int(epochs * args.epochs_scale)), lr=lr)
        if star is not None:
            PINN.evaluate(*star)
    return PINN


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--re", type=float, default=2000)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--hidden", type=int, default=120)
    ap.add_argument("--nf", type=int, default=40000)
    ap.add_argument("--epochs-scale", type=float, default=1.0)
    ap.add_argument("--data", default=None, help="DNS .mat with X_ref,Y_ref,U_ref,V_ref")
    ap.add_argument("--net-params", default=None)
    a = ap.parse_args()
    train(a, net_params=a.net_params)
