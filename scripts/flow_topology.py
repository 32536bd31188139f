#!/usr/bin/env python3
"""Vortex topology of a cavity velocity field: the number behind "Class 1 / Class 2"
(reference README.md:4-9: plain NSFnet at Re = 2000 lands on DNS-like flows AND on a new flow type, "Class 1",
that neither DNS nor ev-NSFnet captures; the README shows streamline pictures only).

Stream function on the DNS grid from (u, v):  u = d(psi)/dy, v = -d(psi)/dx, psi = 0 on the walls.
psi is integrated from the bottom wall with the trapezoid rule, psi(x, y) = int_0^y u dy.  Reported:
  * primary vortex: (x, y) of min psi (the lid moves in +x: the primary vortex turns clockwise, psi < 0) and psi_min,
    with a quadratic sub-cell refinement of the centre;
  * secondary (counter-rotating, psi > 0) eddies: connected regions of psi > eps * |psi_min|, their centre
    (max psi), strength and area fraction, labelled by the corner / wall they sit on;
  * mass-conservation defect: psi at the lid, i.e. int_0^1 u dy per column, which is 0 for a divergence-free field.

    python scripts/flow_topology.py --dns tests/golden/dns/cavity_Re2000_256.mat
    python scripts/flow_topology.py --dns ... --ev-net net.pth [--layers 6 --hidden 80]     (GPU: HIP predict)
    python scripts/flow_topology.py --dns ... --nsfnet-net net.pth --layers 4 --hidden 120
"""
import argparse
import json
import os
import sys

import numpy as np
import scipy.io
import scipy.ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stream_function(X, Y, U):
    """psi(x, y) = int_0^y u dy on a tensor grid (any axis order).  Returns psi with the layout of U."""
    yvar0 = np.abs(np.diff(Y, axis=0)).max() > np.abs(np.diff(Y, axis=1)).max()     # y varies along axis 0?
    ax = 0 if yvar0 else 1
    y = np.moveaxis(Y, ax, 0)
    u = np.moveaxis(U, ax, 0)
    order = np.argsort(y[:, 0])
    y, u = y[order], u[order]
    dy = np.diff(y, axis=0)
    psi = np.zeros_like(u)
    psi[1:] = np.cumsum(0.5 * (u[1:] + u[:-1]) * dy, axis=0)
    inv = np.empty_like(order); inv[order] = np.arange(order.size)
    return np.moveaxis(psi[inv], 0, ax)


def _refine(psi, i, j, X, Y):
    """Quadratic sub-cell estimate of an extremum at grid node (i, j)."""
    def off(a, b, c):
        d = a - 2.0 * b + c
        return 0.0 if d == 0 else float(np.clip(0.5 * (a - c) / d, -0.5, 0.5))
    n0, n1 = psi.shape
    di = off(psi[i - 1, j], psi[i, j], psi[i + 1, j]) if 0 < i < n0 - 1 else 0.0
    dj = off(psi[i, j - 1], psi[i, j], psi[i, j + 1]) if 0 < j < n1 - 1 else 0.0
    i2, j2 = min(max(i + (1 if di > 0 else -1), 0), n0 - 1), min(max(j + (1 if dj > 0 else -1), 0), n1 - 1)
    x = X[i, j] + abs(di) * (X[i2, j] - X[i, j]) + abs(dj) * (X[i, j2] - X[i, j])
    y = Y[i, j] + abs(di) * (Y[i2, j] - Y[i, j]) + abs(dj) * (Y[i, j2] - Y[i, j])
    return float(x), float(y)


def _where(x, y):
    v = "bottom" if y < 0.35 else ("top" if y > 0.65 else "mid")
    h = "left" if x < 0.35 else ("right" if x > 0.65 else "centre")
    return v + "-" + h


def topology(X, Y, U, V, eps=1e-3, min_area=2e-4):
    psi = stream_function(X, Y, U)
    i, j = np.unravel_index(np.argmin(psi), psi.shape)
    cx, cy = _refine(psi, i, j, X, Y)
    psi_min = float(psi[i, j])
    lab, n = scipy.ndimage.label(psi > eps * abs(psi_min))
    eddies = []
    for k in range(1, n + 1):
        m = lab == k
        area = float(m.mean())
        if area < min_area:
            continue
        ii, jj = np.unravel_index(np.argmax(np.where(m, psi, -np.inf)), psi.shape)
        if Y[ii, jj] >= Y.max() - 1e-12:
            continue      # maximum ON the lid: not an eddy but the field's mass defect (psi(lid) != 0) showing through
        ex, ey = _refine(psi, ii, jj, X, Y)
        eddies.append(dict(where=_where(ex, ey), x=ex, y=ey, psi=float(psi[ii, jj]), area=area))
    eddies.sort(key=lambda e: -e["area"])
    # lid row = largest y
    yv0 = np.abs(np.diff(Y, axis=0)).max() > np.abs(np.diff(Y, axis=1)).max()
    top = psi[np.argmax(Y[:, 0]), :] if yv0 else psi[:, np.argmax(Y[0, :])]
    ke = float(np.mean(U ** 2 + V ** 2))
    return dict(primary=dict(x=cx, y=cy, psi_min=psi_min), eddies=eddies,
                mass_defect=float(np.abs(top).max()), kinetic_energy=ke)


def load_dns(path):
    d = scipy.io.loadmat(path)
    return d["X_ref"], d["Y_ref"], d["U_ref"], d["V_ref"]


def predict_field(kind, net_path, X, Y, layers, hidden, evm_path=None, Re=2000.0):
    """(u, v) of a trained net on the DNS grid through the product's HIP value-mode forward (needs the GPU)."""
    import torch
    from nsfnet_amd import engine as eng
    dev = torch.device("cuda:0")
    E = eng.PinnEngine(dev, layers, hidden, Re)
    E.net.load_state_dict(torch.load(net_path, map_location="cpu", weights_only=True))
    u, v, p = E.predict(X.reshape(-1), Y.reshape(-1))
    return u.cpu().numpy().reshape(X.shape).astype(np.float64), v.cpu().numpy().reshape(X.shape).astype(np.float64)


def describe(name, t):
    p = t["primary"]
    s = "%-28s primary vortex (%.4f, %.4f) psi_min %.5f | KE %.4f | mass defect %.1e | eddies: " % (
        name, p["x"], p["y"], p["psi_min"], t["kinetic_energy"], t["mass_defect"])
    return s + ("; ".join("%s (%.3f, %.3f) psi %.2e area %.3f" % (e["where"], e["x"], e["y"], e["psi"], e["area"])
                          for e in t["eddies"]) or "none")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dns", required=True)
    ap.add_argument("--ev-net", default=None)
    ap.add_argument("--nsfnet-net", default=None)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--hidden", type=int, default=None)
    ap.add_argument("--re", type=float, default=2000.0)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    X, Y, U, V = load_dns(a.dns)
    out = {"dns": topology(X, Y, U, V)}
    print(describe("DNS " + os.path.basename(a.dns), out["dns"]))
    for kind, path, dl, dh in (("ev-NSFnet", a.ev_net, 6, 80), ("NSFnet", a.nsfnet_net, 4, 120)):
        if path:
            u, v = predict_field(kind, path, X, Y, a.layers or dl, a.hidden or dh, Re=a.re)
            t = topology(X, Y, u, v)
            t["err_u"] = float(np.linalg.norm(u - U) / np.linalg.norm(U))
            t["err_v"] = float(np.linalg.norm(v - V) / np.linalg.norm(V))
            out[kind] = t
            print(describe("%s %s" % (kind, os.path.basename(path)), t) + " | rel-L2 vs DNS u %.3f v %.3f" % (t["err_u"], t["err_v"]))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
