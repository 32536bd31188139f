#!/usr/bin/env python3
"""Evaluate saved ev-NSFnet checkpoints (<ckpt> and <ckpt>_evm) on a DNS field
(reference: ev-NSFnet/test.py:27-99): percent L2 errors of u, v, p + a .mat of the fields."""
import argparse
import os
import re

import cavity_data as cavity
import pinn_solver as psolver
from config import ConfigManager


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoints", nargs="+")
    ap.add_argument("--data", required=True)
    ap.add_argument("--config", default="configs/production.yaml")
    ap.add_argument("--out", default="./results/test_result")
    a = ap.parse_args()
    cfg = (ConfigManager.from_file(a.config) if os.path.exists(a.config) else ConfigManager()).config
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    for ck in a.checkpoints:
        PINN = psolver.PysicsInformedNeuralNetwork(
            Re=cfg.physics.Re, layers=cfg.network.layers, layers_1=cfg.network.layers_1,
            hidden_size=cfg.network.hidden_size, hidden_size_1=cfg.network.hidden_size_1, N_f=cfg.training.N_f,
            net_params=ck, net_params_1=(ck + "_evm") if os.path.exists(ck + "_evm") else None)
        star = cavity.DataLoader(N_f=cfg.training.N_f, coord_transform=cfg.training.coordinate_transform
                                 ).loading_evaluate_data(a.data)
        m = re.search(r"(\d+)\.pth", ck)
        PINN.evaluate(*star)
        PINN.test(*star, loop=int(m.group(1)) if m else 0, save_dir=a.out)
