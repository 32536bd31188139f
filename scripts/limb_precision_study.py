"""CPU study (numpy, fp64 emulation): what would int8-limb MFMA operands cost in accuracy?

profiles/r03_mfma_dtype_power.txt: at the power limit an i8 MFMA (32x32x32) holds the clock of the bf16 one (32x32x16)
and does twice the MACs - half the energy per MAC.  A product of two 15-bit block-fixed-point operands needs three i8
MFMAs (a1*w1, a1*w0, a0*w1: limbs of 8 + 7 bits) = 1.5 bf16-MFMA equivalents where bf16x3 needs 3.  This script prices
the accuracy of that BEFORE any kernel is written: the 4-stream sweep of oracle/fwdmode_ref.py with every GEMM operand
quantised the way a kernel would (operands first rounded to fp32, products and sums exact - i32 / fp32 accumulation
errors are not modelled, they are the same for every mode), everything elementwise in fp64, against the unquantised fp64
result.  Modes per GEMM family (forward sweep, reverse sweep, dW):
   exact | fp32 (operands rounded to fp32 only) | bf16 | bf16x3 | i8x2 (15-bit rows, 3 terms) | i8x2f (4 terms) | i8x3 (22-bit, 6 terms)
   | i8a3b2 / i8a2b3 (22-bit activations or adjoints x 15-bit weights and the reverse, 5 terms)
Block scales: one per K-vector (forward / reverse sweep: per point and stream over the features, per weight row / column;
dW: per feature over the points of a TILE of `--dw-tile` points, as a kernel accumulating in i32 per tile would have to).

    python scripts/limb_precision_study.py [--n 2048] [--weights seeded|trained]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fwdmode_ref as fr          # noqa: E402  (checker-side script, like fuzz_shapes.py)


def bf16_rne(a):
    u = np.asarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def limbs(M, axis, nl):
    """block-fixed-point along `axis`: value = s * sum_i 128^i q_i, top limb 8 bits signed, lower limbs in [-64, 63]"""
    M = np.asarray(M, np.float32).astype(np.float64)
    top = 127 * 128 ** (nl - 1) + sum(63 * 128 ** i for i in range(nl - 1))
    s = np.max(np.abs(M), axis=axis, keepdims=True) / top
    s[s == 0] = 1.0
    q = np.rint(M / s)
    out = []
    for _ in range(nl - 1):
        hi = np.floor((q + 64) / 128)
        out.append(q - 128 * hi)
        q = hi
    out.append(q)
    return out, s          # out[0] lowest limb


def mm(A, B, mode):
    """A (M, K) @ B (K, N) with both operands quantised as `mode` says"""
    if mode == "exact":
        return A @ B
    A32, B32 = np.asarray(A, np.float32).astype(np.float64), np.asarray(B, np.float32).astype(np.float64)
    if mode == "fp32":
        return A32 @ B32
    if mode in ("bf16", "bf16x3"):
        ah, bh = bf16_rne(A32), bf16_rne(B32)
        if mode == "bf16":
            return ah @ bh
        al, bl = bf16_rne(A32 - ah), bf16_rne(B32 - bh)
        return ah @ bh + al @ bh + ah @ bl
    na, nb, keep = {"i8x2": (2, 2, 1), "i8x2f": (2, 2, 0), "i8x3": (3, 3, 2), "i8a3b2": (3, 2, 1), "i8a2b3": (2, 3, 1)}[mode]
    qa, sa = limbs(A32, 1, na)
    qb, sb = limbs(B32, 0, nb)
    acc = 0.0
    for i in range(na):
        for j in range(nb):
            if i + j >= keep:                                   # drop terms below 128^keep
                acc = acc + (128.0 ** (i + j)) * (qa[i] @ qb[j])
    return acc * sa * sb


def mm_dw(Zs, As, mode, tile):
    """sum over the four streams of Z_s^T (H_out, N) @ A_s (N, H_in): K = points x streams.
    i8x2: block scales per feature, stream and tile of points (one i32 accumulation per stream and tile);
    i8x2jn: ONE i32 accumulation per tile over all four streams, one scale per feature, nothing else;
    i8x2j: ONE i32 accumulation per tile over all four streams - each stream of A is first scaled to unit maximum over
    the tile and the matching stream of Z by the inverse (free: the product is unchanged), then one scale per feature."""
    if mode in ("exact", "fp32", "bf16", "bf16x3"):
        return sum(mm(Zs[s].T, As[s], mode) for s in range(4))
    out = 0.0
    for p0 in range(0, Zs[0].shape[0], tile):
        if mode == "i8x2jn":                                    # joint accumulation WITHOUT the per-stream normalisation
            out = out + mm(np.concatenate([Zs[s][p0:p0 + tile] for s in range(4)], axis=0).T,
                           np.concatenate([As[s][p0:p0 + tile] for s in range(4)], axis=0), "i8x2")
        elif mode == "i8x2j":
            c = [max(float(np.max(np.abs(As[s][p0:p0 + tile]))), 1e-300) for s in range(4)]
            Zt = np.concatenate([Zs[s][p0:p0 + tile] * c[s] for s in range(4)], axis=0)
            At = np.concatenate([As[s][p0:p0 + tile] / c[s] for s in range(4)], axis=0)
            out = out + mm(Zt.T, At, "i8x2")
        else:
            out = out + sum(mm(Zs[s][p0:p0 + tile].T, As[s][p0:p0 + tile], mode) for s in range(4))
    return out


def sweep(params, x, y, Re, fm, bm, dm, tile):
    """pde loss sums and flat gradient (oracle/fwdmode_ref.py forward4 / pde_loss_and_grad / backward4) with quantised GEMMs"""
    x = np.asarray(x, np.float64).reshape(-1); y = np.asarray(y, np.float64).reshape(-1)
    W0, b0 = params[0]
    z = np.outer(x, W0[:, 0]) + np.outer(y, W0[:, 1]) + b0
    zx = np.broadcast_to(W0[:, 0], z.shape).copy(); zy = np.broadcast_to(W0[:, 1], z.shape).copy(); zd = np.zeros_like(z)
    saved, n_lin = [], len(params)
    for l in range(n_lin - 1):
        t = np.tanh(z); d1 = 1.0 - t * t; d2 = -2.0 * t * d1
        saved.append((t, zx, zy, zd))
        a = (t, d1 * zx, d1 * zy, d2 * (zx * zx + zy * zy) + d1 * zd)
        W, b = params[l + 1]
        z, zx, zy, zd = mm(a[0], W.T, fm) + b, mm(a[1], W.T, fm), mm(a[2], W.T, fm), mm(a[3], W.T, fm)
    out = np.stack([z, zx, zy, zd], axis=2)
    eqs = fr.residuals(out, Re)
    N = out.shape[0]
    g = [2.0 * q / N for q in eqs]
    u, v = out[:, 0, 0], out[:, 1, 0]
    adj = np.zeros_like(out)
    adj[:, 0, 0] = g[0] * out[:, 0, 1] + g[1] * out[:, 1, 1]
    adj[:, 1, 0] = g[0] * out[:, 0, 2] + g[1] * out[:, 1, 2]
    adj[:, 0, 1] = g[0] * u + g[2]; adj[:, 0, 2] = g[0] * v
    adj[:, 1, 1] = g[1] * u; adj[:, 1, 2] = g[1] * v + g[2]
    adj[:, 2, 1] = g[0]; adj[:, 2, 2] = g[1]
    adj[:, 0, 3] = -g[0] / Re; adj[:, 1, 3] = -g[1] / Re
    grads = [None] * n_lin
    W, b = params[-1]
    t, zx, zy, zd = saved[-1]
    d1 = 1.0 - t * t; d2 = -2.0 * t * d1
    ap = (t, d1 * zx, d1 * zy, d2 * (zx * zx + zy * zy) + d1 * zd)
    grads[-1] = (mm_dw([adj[:, :, s] for s in range(4)], ap, dm, tile), adj[:, :, 0].sum(axis=0))
    gs = [mm(adj[:, :, s], W, bm) for s in range(4)]
    for l in range(n_lin - 2, -1, -1):
        t, zx, zy, zd = saved[l]
        d1 = 1.0 - t * t; d2 = -2.0 * t * d1; d3 = -2.0 * d1 * (1.0 - 3.0 * t * t)
        ga, gx, gy, gd = gs
        zbs = (d1 * ga + d2 * (zx * gx + zy * gy) + (d3 * (zx * zx + zy * zy) + d2 * zd) * gd,
               d1 * gx + 2.0 * d2 * zx * gd, d1 * gy + 2.0 * d2 * zy * gd, d1 * gd)
        if l == 0:
            grads[0] = (np.stack([x @ zbs[0] + zbs[1].sum(axis=0), y @ zbs[0] + zbs[2].sum(axis=0)], axis=1), zbs[0].sum(axis=0))
        else:
            tp, zxp, zyp, zdp = saved[l - 1]
            d1p = 1.0 - tp * tp; d2p = -2.0 * tp * d1p
            ap = (tp, d1p * zxp, d1p * zyp, d2p * (zxp * zxp + zyp * zyp) + d1p * zdp)
            grads[l] = (mm_dw(zbs, ap, dm, tile), zbs[0].sum(axis=0))
            gs = [mm(zbs[s], params[l][0], bm) for s in range(4)]
    return np.array([float(np.sum(q * q)) for q in eqs]), fr.flatten(grads)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--weights", default="both", choices=("seeded", "trained", "both"))
    ap.add_argument("--dw-tile", type=int, default=64)
    args = ap.parse_args()
    import torch
    L, H, Re = 6, 256, 2000.0
    rng = np.random.RandomState(11)
    N = args.n
    x = np.concatenate([rng.rand(N // 2), np.clip(rng.rand(N // 2) ** 4, 1e-4, 1)]).astype(np.float32)
    y = np.concatenate([rng.rand(N // 2), 1.0 - np.clip(rng.rand(N // 2) ** 4 * 0.2, 1e-4, 1)]).astype(np.float32)
    sets = []
    if args.weights in ("seeded", "both"):
        import bench
        sets.append(("seeded 6x256 (the bench's weights)", bench.seeded_flat(L, H).numpy().astype(np.float64)))
    if args.weights in ("trained", "both"):
        sd = torch.load(os.path.join(ROOT, "tests", "golden", "trained", "ev_re2000_6x256_net.pth"), weights_only=True)
        flat = np.concatenate([np.concatenate([sd["layers.layer_%d.weight" % i].numpy().reshape(-1), sd["layers.layer_%d.bias" % i].numpy().reshape(-1)])
                               for i in range(L + 1)]).astype(np.float64)
        sets.append(("trained 6x256 (Re 2000, 3.3 % vs DNS)", flat))
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    modes = [("fp32", "fp32", "fp32"), ("bf16x3", "bf16x3", "bf16x3"), ("bf16", "bf16", "bf16"),
             ("i8x2", "i8x2", "bf16x3"), ("i8x2f", "i8x2f", "bf16x3"), ("i8x3", "i8x3", "bf16x3"),
             ("i8x2", "bf16x3", "bf16x3"), ("bf16x3", "i8x2", "bf16x3"),
             ("bf16x3", "bf16x3", "i8x2"), ("bf16x3", "bf16x3", "i8x2j"), ("bf16x3", "bf16x3", "i8x2jn"), ("bf16x3", "i8x2", "i8x2"), ("bf16x3", "i8x2", "i8x2j"),
             ("bf16x3", "i8a3b2", "bf16x3"), ("bf16x3", "i8a2b3", "bf16x3"), ("i8a3b2", "bf16x3", "bf16x3"), ("i8a2b3", "bf16x3", "bf16x3"),
             ("bf16x3", "bf16x3", "bf16")]
    for name, flat in sets:
        params = fr.unflatten(flat, 2, 3, L, H)
        s0, g0 = sweep(params, x, y, Re, "exact", "exact", "exact", args.dw_tile)
        print("%s, %d points (half in the lid corners), Re %g: loss sums %s" % (name, N, Re, np.array2string(s0, precision=4)))
        print("  %-8s %-8s %-8s   loss rel err   max eq-sum rel err   gradient rel-L2" % ("forward", "reverse", "dW"))
        for fm, bm, dm in modes:
            s, g = sweep(params, x, y, Re, fm, bm, dm, args.dw_tile)
            print("  %-8s %-8s %-8s   %.2e       %.2e             %.2e" % (fm, bm, dm, abs(s.sum() - s0.sum()) / s0.sum(), np.max(np.abs(s - s0) / s0), rel(g, g0)),
                  flush=True)


if __name__ == "__main__":
    main()
