#!/usr/bin/env python3
"""Headline benchmark: collocation-point NS-residual evals/sec (full training step).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

Workload (BASELINE.json configs[2]): Re=2000 lid-driven cavity, 6x256 tanh
FCNet, 360 000 collocation points PER GPU on a cell-centred uniform grid (weak scaling),
the reference's 2052 boundary points, alpha_b=10, alpha_e=1, Adam lr=1e-3, synthetic
seeded weights.  One "step" = BC forward/backward + 4-stream residual forward + reverse
sweep + weight-gradient GEMMs + gradient reduce (+ one RCCL all-reduce when N > 1) + Adam +
parameter re-layout, i.e. the reference's solve_Adam loop body
(NSFnet/pinn_solver.py:250-254).

Rank 0 prints ONE JSON line; `roofline` is for the dominant kernel, measured live with
HIP events on the launch stream; `cpu_baseline` is the oracle's torch-autograd restatement
of the reference step timed on this host's cores on a bounded point sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X dense MFMA peaks (MI355X_MICROARCH.md, chip table): f32-input 157.3 TFLOP/s, bf16 ~2500 TFLOP/s
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0}
MFMA_PER_PRODUCT = {"fp32": 1, "bf16x3": 3, "bf16": 1}
# HBM bytes per launch of the default workload (6x256, 360k pts) from the PMC passes committed under
# profiles/ (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); None = not measured.
PMC_TRAFFIC_BYTES = {   # profiles/r01_final_bf16x3_pmc_summary.txt, profiles/r01_final_fp32_pmc_summary.txt
    ("bf16x3", "fwd_bf16_kernel"): 8.882e9, ("bf16x3", "bwd_bf16_kernel"): 1.626e10, ("bf16x3", "dw_bf16_kernel"): 1.481e10,
    ("fp32", "fwd_wide_kernel"): 8.885e9, ("fp32", "bwd_wide_kernel"): 1.627e10, ("fp32", "dw_wide_kernel"): 1.481e10,
}
# Matrix-pipe utilisation of the same launches, SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8), from the
# same PMC summaries (the chip holds 1.6-1.9 GHz under this load, so this is not frac x mfma_per_product).
PMC_MFMA_BUSY = {
    ("bf16x3", "fwd_bf16_kernel"): 0.52, ("bf16x3", "bwd_bf16_kernel"): 0.38, ("bf16x3", "dw_bf16_kernel"): 0.57,
    ("fp32", "fwd_wide_kernel"): 0.80, ("fp32", "bwd_wide_kernel"): 0.66, ("fp32", "dw_wide_kernel"): 0.83,
}


def weight_count(L, H, n_out=3):
    return 2 * H + (L - 1) * H * H + n_out * H


def cavity_boundary(nx=513):
    s = np.linspace(0.0, 1.0, nx)
    lid = 1.0 - np.cosh(10.0 * (s - 0.5)) / np.cosh(5.0)
    z, o = np.zeros(nx), np.ones(nx)
    return (np.concatenate([s, s, z, o]), np.concatenate([z, o, s, s]),
            np.concatenate([z, lid, z, z]), np.zeros(4 * nx))


def grid_block(nx_local, ny, rank, world):
    """Rows [rank*nx_local, (rank+1)*nx_local) of the (world*nx_local) x ny cell-centred grid."""
    nx = nx_local * world
    xs = (np.arange(rank * nx_local, (rank + 1) * nx_local) + 0.5) / nx
    ys = (np.arange(ny) + 0.5) / ny
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    return X.reshape(-1).astype(np.float32), Y.reshape(-1).astype(np.float32)


def seeded_flat(L, H, n_out=3, seed=1234):
    """Default nn.Linear init of the reference FCNet under torch.manual_seed(seed), flattened
    in state_dict order (SURVEY.md 8d synthetic weights)."""
    torch.manual_seed(seed)
    widths = [2] + [H] * L + [n_out]
    parts = []
    for i in range(len(widths) - 1):
        lin = torch.nn.Linear(widths[i], widths[i + 1])
        parts += [lin.weight.detach().reshape(-1), lin.bias.detach().reshape(-1)]
    return torch.cat(parts)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


T0 = time.perf_counter()


def time_kernel(fn, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))   # ms


def cpu_baseline(L, H, Re, n_sample, steps=12):
    from oracle import autograd_ref as ar
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, 16))      # the GPU box gives one GPU a 16-core share; more threads only oversubscribe
    torch.set_num_threads(cores)
    net = ar.seeded_net(3, L, H, seed=1234)
    o = ar.NSFnetOracle(net, Re, alpha_b=10.0, alpha_e=1.0, lr=1e-3)
    side = int(round(n_sample ** 0.5))
    x, y = ar.uniform_grid(side, side)
    o.set_data(x, y, *ar.cavity_boundary())
    log("cpu_baseline: warm-up step (%d pts, %d threads)" % (side * side, cores))
    o.step()                      # warm-up
    log("cpu_baseline: timing %d steps" % steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        o.step()
    dt = (time.perf_counter() - t0) / steps
    n = side * side
    return dict(value=n / dt, unit="collocation-pt residual evals/s", cores=cores, kind="port",
                sample="%d-pt uniform grid (%dx%d), 6x256, 1 warm-up + %d timed full steps of the torch-autograd "
                       "restatement (oracle/autograd_ref.py), %.2f s/step" % (n, side, side, steps, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--grid", type=int, default=600, help="per-GPU collocation grid is grid x grid")
    ap.add_argument("--re", type=float, default=2000.0)
    ap.add_argument("--precision", default=os.environ.get("NSFNET_PRECISION", "bf16x3"),
                    help="bf16x3 (default: bf16 MFMA, hi/lo split, meets the 1e-4 loss-parity bar) | fp32 "
                         "(f32-input MFMA, bit-exact fp32) | bf16 (plain bf16 operands, fast mode, no parity claim)")
    ap.add_argument("--alt-precision", default="fp32", help="second mode reported in the 'alt' block ('' = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=16384)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)      # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm.  NSFNET_DIST_BACKEND=gloo only exists to rehearse the rank logic on a
        # box with fewer GPUs than ranks (RCCL refuses two ranks on one device).
        backend = os.environ.get("NSFNET_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        pg = dist.group.WORLD

    from nsfnet_amd import engine as eng

    L, H, Re = args.layers, args.hidden, args.re
    n_local = args.grid * args.grid
    n_global = n_local * world
    E = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0, process_group=pg, world_size=world,
                       precision=args.precision)
    E.net.set_flat(seeded_flat(L, H))
    x, y = grid_block(args.grid, args.grid, rank, world)
    E.set_collocation(x, y, n_global=n_global)
    xb, yb, ub, vb = cavity_boundary()
    nb = xb.size
    per = nb // world                 # reference split: contiguous blocks, last rank takes the remainder
    lo, hi = rank * per, (nb if rank == world - 1 else (rank + 1) * per)
    E.set_boundary(xb[lo:hi], yb[lo:hi], ub[lo:hi], vb[lo:hi], n_global=nb)
    lr = 1e-3

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    log("setup done: %d pts/GPU, world %d" % (n_local, world))
    for i in range(args.warmup):
        E.step(lr)
        if i == 0:
            torch.cuda.synchronize(); log("first step done")
    barrier()
    log("warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        E.step(lr)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(E.loss_terms()["loss"])
    log("timed region done: %.2f ms/step, loss %.6f" % (1e3 * dt / args.steps, loss))

    def kernel_report(E_, prec, ms_step):
        f = E_.plan_f
        n_launch = n_local
        if hasattr(f, "chunks"):          # $NSFNET_CHUNK_POINTS: time one pass (the first, largest chunk)
            f = f.chunks[0]
            n_launch = f.n
        c = 2.0 / n_global
        reps = max(3, min(10, args.steps))
        t_fwd = time_kernel(lambda: f.forward(Re, save=True), reps)
        t_bwd = time_kernel(lambda: f.backward(Re, (c, c, c, 0.0), phases=1), reps)
        t_dw = time_kernel(lambda: f.backward(Re, (c, c, c, 0.0), phases=2), reps)
        log("[%s] kernel ms: fwd %.3f bwd %.3f dw %.3f" % (prec, t_fwd, t_bwd, t_dw))
        pw = weight_count(L, H)
        flops_each = 8.0 * pw * n_launch       # fwd, dX sweep and dW GEMM each carry 2*4*P_w FLOP per point
        wide = H > 256 or (H > 224 and prec == "fp32")      # 64-column-tile kernels (csrc/capi.hip pick_wide)
        names = {"fp32": ("fwd_wide_kernel", "bwd_wide_kernel", "dw_wide_kernel") if wide else ("fwd_kernel", "bwd_kernel", "dw_kernel"),
                 "bf16x3": ("fwd_bf16_kernel", "bwd_bf16_kernel", "dw_bf16_kernel"),
                 "bf16": ("fwd_bf16_kernel", "bwd_bf16_kernel", "dw_bf16_kernel")}[prec]
        if H > 256 and prec != "fp32":     # wide nets: 64 features per wave, blocked dW
            names = ("fwd_bf16_wide_kernel", "bwd_bf16_wide_kernel", "dw_bf16_wide_kernel")
        kernels = dict(zip(names, (t_fwd, t_bwd, t_dw)))
        dom = max(kernels, key=kernels.get)
        achieved = flops_each / (kernels[dom] * 1e-3) / 1e12
        dom_prec = prec
        peak = MFMA_PEAK_TFLOPS[dom_prec]
        default_cfg = (L, H, args.grid, n_launch) == (6, 256, 600, n_local)
        traffic = PMC_TRAFFIC_BYTES.get((prec, dom)) if default_cfg else None
        pipe_busy = PMC_MFMA_BUSY.get((prec, dom)) if default_cfg else None
        return dict(bound="mfma", kernel=dom, achieved=achieved, peak=peak, unit="TFLOP/s", frac=achieved / peak,
                    traffic=traffic, mfma_per_product=MFMA_PER_PRODUCT[dom_prec],
                    mfma_issue_frac=achieved * MFMA_PER_PRODUCT[dom_prec] / peak, matrix_pipe_busy_pmc=pipe_busy,
                    kernel_ms={k: round(v, 4) for k, v in kernels.items()},
                    step_tflops=24.0 * pw * n_local / (ms_step * 1e-3) / 1e12,
                    forward_only_evals_per_s=n_launch / (t_fwd * 1e-3))

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = n_global * args.steps / dt
        # ---- per-kernel timing of the three MFMA kernels (outside the timed region) ----
        prec = args.precision if args.precision in MFMA_PEAK_TFLOPS else "fp32"
        roofline = kernel_report(E, prec, ms_per_step)
        out = dict(metric="collocation-pt NS-residual evals/sec, Re=2000 6x256 MLP",
                   value=value, unit="collocation-pt residual evals/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype={"fp32": "f32", "bf16x3": "bf16x3 (bf16 MFMA, hi/lo split, f32 accumulate)",
                                            "bf16": "bf16"}.get(prec, prec), data="synthetic",
                   config=dict(workload="Re=%g cavity, %dx%d tanh FCNet, %d collocation pts/GPU (%dx%d cell-centred "
                                        "uniform grid) + 2052 BC pts, full Adam step, precision %s"
                                        % (Re, L, H, n_local, args.grid, args.grid, args.precision),
                               global_points=n_global, parallelism="dp%d" % world, final_loss=loss),
                   roofline=roofline)
        if args.alt_precision and args.alt_precision != args.precision and world == 1:
            del E
            torch.cuda.empty_cache()
            E2 = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0, precision=args.alt_precision)
            E2.net.set_flat(seeded_flat(L, H))
            E2.set_collocation(x, y, n_global=n_global)
            E2.set_boundary(xb[lo:hi], yb[lo:hi], ub[lo:hi], vb[lo:hi], n_global=nb)
            for _ in range(2):
                E2.step(lr)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            k2 = max(3, args.steps // 2)
            for _ in range(k2):
                E2.step(lr)
            torch.cuda.synchronize()
            ms2 = 1e3 * (time.perf_counter() - t1) / k2
            out["alt"] = dict(precision=args.alt_precision, value=n_global / (ms2 * 1e-3), ms_per_step=ms2, steps=k2,
                              roofline=kernel_report(E2, args.alt_precision, ms2))
            del E2
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(L, H, Re, args.cpu_sample)
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
