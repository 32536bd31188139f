#!/usr/bin/env python3
"""Headline benchmark: collocation-point NS-residual evals/sec (full training step).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1..5]

N > 1: one rank per GPU over RCCL.  Started either by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the environment) or plainly as above - then this process starts the N ranks itself as CHILD
processes through torch.distributed.run on 127.0.0.1 (never exec), forwards rank 0's JSON line and exits with
the children's code.

Workloads = BASELINE.json `configs`, synthetic cell-centred uniform grids, the reference's 2052
boundary points, alpha_b = 10, alpha_e = 1, Adam lr = 1e-3, seeded weights.  --scaling weak (default): the PER-GPU
share below is fixed; --scaling strong: the TOTAL is fixed at 8 x that share (config 3: 2.88 M points)
and split in equal row blocks over the N ranks (N must divide 8 x nx):
  --config 3 (default)  Re=2000 NSFnet, 6x256, 600x600 = 360 000 pts/GPU, bf16x3   <- the metric's configuration
  --config 2            Re=1000 NSFnet, 6x128, 400x300 = 120 000 pts/GPU, fp32
  --config 4            ev-NSFnet Re=4000, 6x256 + 4x40 entropy net, 250x1000 = 250 000 pts/GPU (1 M over 4 GPUs)
  --config 5            ev-NSFnet Re=10000, 8x400 + 4x40, 250x2000 = 500 000 pts/GPU (4 M over 8 GPUs)
  --config 1            Re=100 NSFnet, 4x50, 100x100 (the reference's CPU-runnable case)
One "step" = BC forward/backward + 4-stream residual forward + reverse sweep + weight-gradient GEMMs + gradient
reduce (+ ONE RCCL all-reduce when N > 1) + Adam + parameter re-layout, i.e. the reference's solve_Adam loop
body (NSFnet/pinn_solver.py:250-254, ev-NSFnet/pinn_solver.py:456-472).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, timed live with HIP events on the launch
stream; its `traffic` / `matrix_pipe_busy_pmc` come from profiles/r03_pmc.json (written by
scripts/pmc_summarize.py from separate rocprofv3 --pmc passes) and are null unless that file was measured on
exactly this csrc/ (content hash) and this shape.  `sustained` = >= 2 s of back-to-back steps after the timed
region (clock-settled figure).  `cpu_baseline` = the oracle's torch-autograd restatement of the reference step
on this host's cores, bounded sample.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X dense MFMA peaks (MI355X_MICROARCH.md, chip table): f32-input 157.3 TFLOP/s, bf16 ~2500 TFLOP/s
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md
MFMA_PER_PRODUCT = {"fp32": 1, "bf16x3": 3, "bf16": 1}
PMC_JSON = os.path.join(ROOT, "profiles", "r03_pmc.json")
STRONG_SHARES = 8      # --scaling strong: total = this many per-GPU shares of the config (2.88 M = 8 x 360 000)

CONFIGS = {   # flavour, Re, L, H, (nx_local, ny), precision, nominal GPU count of the BASELINE config
    1: dict(flavour="nsfnet", re=100.0, layers=4, hidden=50, grid=(100, 100), precision="bf16x3", gpus=1),
    2: dict(flavour="nsfnet", re=1000.0, layers=6, hidden=128, grid=(400, 300), precision="fp32", gpus=1),
    3: dict(flavour="nsfnet", re=2000.0, layers=6, hidden=256, grid=(600, 600), precision="bf16x3", gpus=1),
    4: dict(flavour="ev", re=4000.0, layers=6, hidden=256, grid=(250, 1000), precision="bf16x3", gpus=4),
    5: dict(flavour="ev", re=10000.0, layers=8, hidden=400, grid=(250, 2000), precision="bf16x3", gpus=8),
}
EV_NET = (4, 40)          # entropy net of ev-NSFnet/config.py:17-22
ALPHA_EVM = 0.05


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


def local_rows(nx, scaling, world):
    """Rows of the global grid one rank holds: weak = the config's per-GPU share, strong = an equal part of
    STRONG_SHARES shares (config 3's net on 2.88 M points in total)."""
    if scaling == "weak":
        return nx
    total = nx * STRONG_SHARES
    if total % world:
        raise SystemExit("bench.py: --scaling strong needs the rank count (%d) to divide %d grid rows" % (world, total))
    return total // world


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


T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


def csrc_hash():
    """Content hash of the kernel sources: PMC figures are only quoted for the build they were measured on."""
    d = os.path.join(ROOT, "nsfnet_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_lookup(kernel, prec, L, H, points):
    """(traffic bytes per launch, matrix-pipe busy fraction, source, clock held in GHz) from profiles/r03_pmc.json or Nones."""
    try:
        doc = json.load(open(PMC_JSON))
    except Exception:
        return None, None, None, None
    if doc.get("csrc_hash") != csrc_hash():
        return None, None, "profiles/r03_pmc.json was measured on another csrc/ (%s)" % doc.get("csrc_hash"), None
    for e in doc.get("entries", []):
        if (e["kernel_short"], e["precision"], e["layers"], e["hidden"], e["points"]) == (kernel, prec, L, H, points):
            return e.get("traffic_bytes"), e.get("mfma_busy"), "profiles/r03_pmc.json:" + e["kernel"], e.get("clock_ghz")
    return None, None, None, None


def time_kernel(fn, reps):
    fn(); fn()                      # (steady state: the first launch after a pause pays the clock ramp)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))   # ms


def cpu_baseline(flavour, L, H, Re, n_sample, steps=12):
    from oracle import autograd_ref as ar
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, 16))      # the GPU box gives one GPU a 16-core share; more threads only oversubscribe
    torch.set_num_threads(cores)
    net = ar.seeded_net(3, L, H, seed=1234)
    side = max(8, int(round(n_sample ** 0.5)))
    x, y = ar.uniform_grid(side, side)
    if flavour == "ev":
        net_e = ar.seeded_net(1, EV_NET[0], EV_NET[1], seed=4321)
        o = ar.EvNSFnetOracle(net, net_e, Re, ALPHA_EVM, alpha_b=10.0, alpha_e=1.0, lr=1e-3)
    else:
        o = ar.NSFnetOracle(net, Re, alpha_b=10.0, alpha_e=1.0, lr=1e-3)
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
                sample="%d-pt uniform grid (%dx%d), %s %dx%d, 1 warm-up + %d timed full steps of the torch-autograd "
                       "restatement (oracle/autograd_ref.py), %.2f s/step" % (n, side, side, flavour, L, H, steps, dt))


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run,
    rendezvous on 127.0.0.1), pass rank 0's JSON line through, return the children's exit code.  Runs before
    anything in this process touches the GPU; the parent never execs."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    # --standalone: the agent binds its own free rendezvous port (no pick-then-close race with other jobs of the host)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting %d ranks: %s" % (n, " ".join(cmd[1:])))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = []
    for line in r.stdout.splitlines():
        try:
            if isinstance(json.loads(line), dict):
                lines.append(line)
                continue
        except ValueError:
            pass
        print(line, file=sys.stderr)
    if r.returncode == 0 and len(lines) != 1:
        print("bench.py: expected one JSON line from rank 0, got %d" % len(lines), file=sys.stderr)
        return 1
    for line in lines:
        print(line, flush=True)
    return r.returncode


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json config number (1-based)")
    ap.add_argument("--flavour", choices=("nsfnet", "ev"), default=None)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--hidden", type=int, default=None)
    ap.add_argument("--grid", default=None, help="per-GPU collocation grid: G (= GxG) or AxB")
    ap.add_argument("--re", type=float, default=None)
    ap.add_argument("--precision", default=os.environ.get("NSFNET_PRECISION"),
                    help="bf16x3 (bf16 MFMA, hi/lo split, meets the 1e-4 loss-parity bar) | fp32 (f32-input MFMA, "
                         "bit-exact fp32) | bf16 (plain bf16 operands, fast mode, no parity claim); default per config")
    ap.add_argument("--alt-precision", default=None, help="other modes reported in the 'alts' list, comma-separated ('' = "
                                                            "skip; default: fp32,bf16 for config 3 at N = 1); 'alt' = the first")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: the config's per-GPU share on every rank; strong: 8 shares in total, split over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=None)
    ap.add_argument("--sustain-seconds", type=float, default=2.0)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.flavour = args.flavour or cfg["flavour"]
    args.layers = args.layers or cfg["layers"]
    args.hidden = args.hidden or cfg["hidden"]
    args.re = args.re or cfg["re"]
    args.precision = args.precision or cfg["precision"]
    if args.grid is None:
        args.nx, args.ny = cfg["grid"]
    else:
        g = [int(v) for v in str(args.grid).lower().split("x")]
        args.nx, args.ny = (g[0], g[0]) if len(g) == 1 else (g[0], g[1])
    custom = (args.flavour, args.layers, args.hidden, (args.nx, args.ny), args.re) != (
        cfg["flavour"], cfg["layers"], cfg["hidden"], cfg["grid"], cfg["re"])
    args.config_name = "custom" if custom else "BASELINE configs[%d]" % (args.config - 1)
    if args.alt_precision is None:
        args.alt_precision = "fp32,bf16" if (args.config == 3 and not custom and args.scaling == "weak") else ""
    args.alt_list = [p for p in args.alt_precision.split(",") if p and p != args.precision]
    if args.cpu_sample is None:     # ~10-20 s of CPU work whatever the net: scale the sample with 1 / P_w
        args.cpu_sample = max(1024, int(16384 * weight_count(6, 256) / weight_count(args.layers, args.hidden)))
        args.cpu_sample = min(args.cpu_sample, 65536)
    return args


def build_engine(eng, dev, args, precision, pg, world, rank):
    L, H, Re = args.layers, args.hidden, args.re
    ev = args.flavour == "ev"
    E = eng.PinnEngine(dev, L, H, Re, alpha_b=10.0, alpha_e=1.0, process_group=pg, world_size=world,
                       precision=precision, flavour=args.flavour, n_hidden_e=EV_NET[0] if ev else None,
                       hidden_e=EV_NET[1] if ev else None, alpha_evm=ALPHA_EVM if ev else 0.0)
    E.net.set_flat(seeded_flat(L, H))
    if ev:
        E.net_e.set_flat(seeded_flat(EV_NET[0], EV_NET[1], n_out=1, seed=4321))
    nxl = local_rows(args.nx, args.scaling, world)
    x, y = grid_block(nxl, args.ny, rank, world)
    n_local = nxl * args.ny
    E.set_collocation(x, y, n_global=n_local * world)
    xb, yb, ub, vb = cavity_boundary()
    nb = xb.size
    per = nb // world                 # reference split: contiguous blocks, last rank takes the remainder
    lo, hi = rank * per, (nb if rank == world - 1 else (rank + 1) * per)
    E.set_boundary(xb[lo:hi], yb[lo:hi], ub[lo:hi], vb[lo:hi], n_global=nb)
    return E


def kernel_report(E_, args, prec, ms_step, n_local, n_global):
    L, H, Re = args.layers, args.hidden, args.re
    f = E_.plan_f
    n_launch = n_local
    if hasattr(f, "chunks"):          # $NSFNET_CHUNK_POINTS: time one pass (the first, largest chunk)
        f = f.chunks[0]
        n_launch = f.n
    c = 2.0 / n_global
    ev = E_.net_e is not None
    e = E_.plan_e.pred[0][:n_launch] if ev else None
    kw = dict(e=e, vis_t0=E_.vis_t0, alpha_evm=E_.alpha_evm) if ev else {}
    coef = (c, c, c, 0.1 * c if ev else 0.0)
    reps = max(3, min(10, args.steps))
    t_fwd = time_kernel(lambda: f.forward(Re, save=True, **kw), reps)
    t_bwd = time_kernel(lambda: f.backward(Re, coef, e=e, phases=1), reps)
    t_dw = time_kernel(lambda: f.backward(Re, coef, e=e, phases=2), reps)
    log("[%s] kernel ms: fwd %.3f bwd %.3f dw %.3f" % (prec, t_fwd, t_bwd, t_dw))
    pw = weight_count(L, H)
    flops_each = 8.0 * pw * n_launch       # fwd, dX sweep and dW GEMM each carry 2*4*P_w FLOP per point
    kernels = dict(zip(f.kernel_names(), (t_fwd, t_bwd, t_dw)))
    dom = max(kernels, key=kernels.get)
    achieved = flops_each / (kernels[dom] * 1e-3) / 1e12
    peak = MFMA_PEAK_TFLOPS[prec]
    traffic, pipe_busy, src, clock = pmc_lookup(dom, prec, L, H, n_launch)
    step_flops = 24.0 * pw * n_local + (2.0 * weight_count(EV_NET[0], EV_NET[1], 1) * n_local if ev else 0.0)
    # second reading of the same launch: the spill traffic (PMC) against the HBM peak - what DESIGN.md 4.3 shows the
    # bf16x3 sweeps are actually limited by (null without a PMC record of this build)
    hbm = None if not traffic else dict(achieved=traffic / (kernels[dom] * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                        frac=traffic / (kernels[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS)
    return dict(bound="mfma", kernel=dom, achieved=achieved, peak=peak, unit="TFLOP/s", frac=achieved / peak,
                traffic=traffic, hbm=hbm, pmc_source=src, mfma_per_product=MFMA_PER_PRODUCT[prec],
                mfma_issue_frac=achieved * MFMA_PER_PRODUCT[prec] / peak, matrix_pipe_busy_pmc=pipe_busy,
                # clock the kernel held in the PMC pass (GRBM_GUI_ACTIVE / 8 / duration; 2.4 GHz nominal): the bf16x3 kernels
                # of the headline shape are held at ~1.7 GHz by board power
                clock_ghz_pmc=clock,
                algorithmic_flop_per_launch=flops_each,
                kernel_ms={k: round(v, 4) for k, v in kernels.items()},
                step_tflops=step_flops / (ms_step * 1e-3) / 1e12,
                forward_only_evals_per_s=n_launch / (t_fwd * 1e-3))


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))        # (nothing above has touched the GPU)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)      # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = None
    backend = "none"
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
    nxl = local_rows(args.nx, args.scaling, world)
    n_local = nxl * args.ny
    n_global = n_local * world
    E = build_engine(eng, dev, args, args.precision, pg, world, rank)
    lr = 1e-3

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    log("%s: %s %dx%d, %d pts/GPU (%s scaling), world %d, %s" % (args.config_name, args.flavour, L, H, n_local, args.scaling, world, args.precision))
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
    log("timed region done: %.3f ms/step, loss %.6f" % (1e3 * dt / args.steps, loss))

    # ---- sustained: >= sustain-seconds of back-to-back steps (every rank takes part: the step all-reduces) ----
    sustained = None
    if args.sustain_seconds > 0:
        chunk = max(10, int(0.25 / max(dt / args.steps, 1e-5)))
        n_sus = int(np.ceil(args.sustain_seconds / max(dt / args.steps, 1e-5) / chunk)) * chunk   # same count on every rank
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_sus):
            E.step(lr)
        barrier()
        ds = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([ds], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            ds = float(t.item())
        sustained = dict(seconds=ds, steps=n_sus, ms_per_step=1e3 * ds / n_sus, value=n_global * n_sus / ds)
        log("sustained: %d steps in %.2f s = %.3f ms/step" % (n_sus, ds, 1e3 * ds / n_sus))

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = n_global * args.steps / dt
        prec = args.precision
        roofline = kernel_report(E, args, prec, ms_per_step, n_local, n_global)
        ev = args.flavour == "ev"
        out = dict(metric="collocation-pt NS-residual evals/sec, Re=%g %dx%d MLP" % (Re, L, H),
                   value=value, unit="collocation-pt residual evals/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling=args.scaling,
                   vs_baseline=None, dtype={"fp32": "f32", "bf16x3": "bf16x3 (bf16 MFMA, hi/lo split, f32 accumulate)",
                                            "bf16": "bf16"}.get(prec, prec), data="synthetic",
                   config=dict(workload="%s: %sRe=%g cavity, %dx%d tanh FCNet%s, %d collocation pts/GPU (%dx%d block of "
                                        "the cell-centred uniform grid) + 2052 BC pts, full Adam step, precision %s"
                                        % (args.config_name, "ev-NSFnet " if ev else "", Re, L, H,
                                           " + %dx%d entropy net" % EV_NET if ev else "", n_local, nxl, args.ny, prec),
                               global_points=n_global, parallelism="dp%d" % world, final_loss=loss,
                               # what moved the per-step all-reduce: "rccl" (torch backend "nccl" on ROCm) is the product
                               # path; "gloo" only ever appears in single-GPU rehearsals of the rank logic
                               backend={"nccl": "rccl"}.get(backend, backend),
                               peak_device_mem_gb=round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
                               csrc_hash=csrc_hash()),
                   roofline=roofline, sustained=sustained)
        if args.alt_list and world == 1:
            del E
            out["alts"] = []
            for alt in args.alt_list:
                torch.cuda.empty_cache()
                E2 = build_engine(eng, dev, args, alt, None, 1, 0)
                for _ in range(3):
                    E2.step(lr)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                k2 = max(3, args.steps // 4)
                for _ in range(k2):
                    E2.step(lr)
                torch.cuda.synchronize()
                ms2 = 1e3 * (time.perf_counter() - t1) / k2
                out["alts"].append(dict(precision=alt, value=n_global / (ms2 * 1e-3), ms_per_step=ms2, steps=k2,
                                        final_loss=float(E2.loss_terms()["loss"]),
                                        roofline=kernel_report(E2, args, alt, ms2, n_local, n_global)))
                del E2
            torch.cuda.empty_cache()
            out["alt"] = out["alts"][0]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.flavour, L, H, Re, args.cpu_sample)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
