#!/usr/bin/env python3
"""ev-NSFnet staged training (reference flow: ev-NSFnet/train.py:74-224) on the MI355X engine.

    torchrun --nproc_per_node=N train.py --config configs/production.yaml [--dry-run] [--epochs-scale s]

One process per GPU; torchrun's RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* contract; backend "nccl"
(= RCCL on ROCm).  Every rank holds a contiguous shard of the points; one all-reduce per step."""
import argparse
import os

import numpy as np
import torch
import torch.distributed as dist

import cavity_data as cavity
import pinn_solver as psolver
from config import ConfigManager
from logger import get_logger


def setup_distributed():
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) < 2:
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    dist.init_process_group(backend=os.environ.get("NSFNET_DIST_BACKEND", "nccl"))
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="configs/production.yaml")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--epochs-scale", type=float, default=1.0)
    ap.add_argument("--data", default=None, help="DNS .mat (X_ref,Y_ref,U_ref,V_ref,P_ref)")
    args = ap.parse_args()
    mgr = ConfigManager.from_file(args.config) if os.path.exists(args.config) else ConfigManager()
    cfg = mgr.config
    distributed = setup_distributed()
    rank = int(os.environ["RANK"])
    log = get_logger(cfg.experiment_name, rank=rank)
    if rank == 0:
        log.header("configuration")
        mgr.print_config()
        for i, st in enumerate(cfg.training.training_stages, 1):
            log.info("%02d | %-8s | alpha=%.3g | epochs=%s | lr=%.2e" % (i, st.name, st.alpha, format(st.epochs, ","), st.lr))
    if args.dry_run:
        return
    try:
        PINN = psolver.PysicsInformedNeuralNetwork(
            Re=cfg.physics.Re, layers=cfg.network.layers, layers_1=cfg.network.layers_1,
            hidden_size=cfg.network.hidden_size, hidden_size_1=cfg.network.hidden_size_1, N_f=cfg.training.N_f,
            alpha_evm=cfg.physics.alpha_evm, bc_weight=cfg.physics.bc_weight, eq_weight=cfg.physics.eq_weight,
            supervised_data_weight=cfg.supervision.loss_weight if cfg.supervision.enabled else 0.0)
        PINN.log_interval = cfg.training.log_interval
        loader = cavity.DataLoader(path="./datasets/", N_f=cfg.training.N_f, N_b=1000,
                                   sort_training_points=cfg.training.sort_training_points,
                                   sdf_weighting=cfg.training.sdf_weighting,
                                   coord_transform=cfg.training.coordinate_transform)
        PINN.set_boundary_data(X=loader.loading_boundary_data())
        if distributed:   # every rank must shard the SAME point set: rank 0 samples, the others receive
            pts = [loader.loading_training_data() + (loader.get_sdf_weights(),)] if rank == 0 else [None]
            dist.broadcast_object_list(pts, src=0)
            xf, yf, sdf = pts[0]
        else:
            xf, yf = loader.loading_training_data()
            sdf = loader.get_sdf_weights()
        PINN.set_coordinate_transform(loader.get_coord_scale())
        PINN.set_eq_training_data(X=(xf, yf), weights=sdf)
        ref = args.data or "./data/cavity_Re%s_256_Uniform.mat" % cfg.physics.Re
        star = loader.loading_evaluate_data(ref) if os.path.exists(ref) else None
        sup = cfg.supervision
        if sup.enabled and sup.num_samples > 0 and star is not None:
            n = min(int(sup.num_samples), star[0].shape[0])
            idx = np.random.default_rng(0).choice(star[0].shape[0], size=n, replace=False)   # same on every rank
            PINN.set_supervised_data(tuple(a[idx] for a in star))
            PINN.set_supervised_loss_weight(sup.loss_weight)
        else:
            PINN.clear_supervised_data()
            PINN.set_supervised_loss_weight(0.0)
        for st in cfg.training.training_stages:
            if rank == 0:
                log.stage(st.name, st.alpha, st.epochs, st.lr)
            PINN.current_stage = st.name
            PINN.set_alpha_evm(st.alpha)
            PINN.train(num_epoch=max(1, int(st.epochs * args.epochs_scale)), lr=st.lr)
            if rank == 0 and star is not None:
                PINN.evaluate(*star)
        if rank == 0:
            log.header("training completed")
    finally:
        if distributed and dist.is_initialized():
            dist.destroy_process_group()
        log.close()


if __name__ == "__main__":
    main()
