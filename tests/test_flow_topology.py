"""Row g of the grading table: the converged field and the "Class 1 / Class 2" split of the reference's README
(README.md:4-9), as numbers instead of streamline pictures (scripts/flow_topology.py).

Class 2 = the DNS-like flow (what ev-NSFnet finds): anchored here on the DNS files the reference ships.
Class 1 = "a new flow type that is not captured by DNS or by ev-NSFnet", found by plain NSFnet for some
initialisations.  PARITY UNPINNED: the reference ships no Class-1 field, weights or numbers (only the picture
resources/ev_NSFnet.png; the GIFs are in .MISSING_LARGE_BLOBS), so what is asserted for the plain-NSFnet end state
of THIS engine (tests/golden/trained/nsfnet_re2000_4x120_net.pth: the reference's own 1.6 M-step schedule,
scripts/converge_nsfnet.py, profiles/r02_convergence_nsfnet_re2000*.json*) is what can be asserted without one:
it is a genuine steady Navier-Stokes solution of the same boundary-value problem (PDE residuals <= 1e-5, lid and
walls matched, divergence-free) with ONE closed primary vortex whose topology differs from DNS - not a garbage field.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import flow_topology as ft        # noqa: E402

DNS = os.path.join(HERE, "golden", "dns")


def test_dns_topology_is_the_known_cavity_flow():
    """CPU: the metric itself, on the reference's DNS fields - one clockwise primary vortex near the centre that
    drifts towards it with Re, bottom-right and bottom-left corner eddies that grow with Re, a top-left eddy from
    Re = 3000 on (the textbook lid-driven-cavity sequence), and a mass defect at the level of the file's accuracy."""
    t = {Re: ft.topology(*ft.load_dns(os.path.join(DNS, f))) for Re, f in
         ((2000, "cavity_Re2000_256.mat"), (3000, "cavity_Re3000_256_Uniform.mat"), (5000, "cavity_Re5000_256_Uniform.mat"))}
    for Re, r in t.items():
        p = r["primary"]
        assert 0.50 < p["x"] < 0.54 and 0.52 < p["y"] < 0.56 and -0.125 < p["psi_min"] < -0.10, (Re, p)
        where = [e["where"] for e in r["eddies"]]
        assert where[:2] == ["bottom-right", "bottom-left"], (Re, where)
        assert r["mass_defect"] < 1e-3
    assert t[2000]["primary"]["x"] > t[3000]["primary"]["x"] > t[5000]["primary"]["x"]
    br = [t[Re]["eddies"][0]["area"] for Re in (2000, 3000, 5000)]
    assert br[0] < br[1] < br[2]
    assert any(e["where"] == "top-left" and e["area"] > 0.004 for e in t[3000]["eddies"])
    assert any(e["where"] == "top-left" and e["area"] > 0.01 for e in t[5000]["eddies"])


def test_stream_function_of_an_analytic_vortex():
    """psi = -sin^2(pi x) sin^2(pi y): u = psi_y, v = -psi_x; the recovered centre, strength and (absent) eddies."""
    n = 201
    s = np.linspace(0.0, 1.0, n)
    X, Y = np.meshgrid(s, s)               # y along axis 0, as in the DNS files
    U = -np.sin(np.pi * X) ** 2 * 2 * np.pi * np.sin(np.pi * Y) * np.cos(np.pi * Y)
    V = 2 * np.pi * np.sin(np.pi * X) * np.cos(np.pi * X) * np.sin(np.pi * Y) ** 2
    for Xg, Yg, Ug, Vg in ((X, Y, U, V), (X.T, Y.T, U.T, V.T)):        # either axis order
        t = ft.topology(Xg, Yg, Ug, Vg)
        assert abs(t["primary"]["x"] - 0.5) < 2e-3 and abs(t["primary"]["y"] - 0.5) < 2e-3
        assert abs(t["primary"]["psi_min"] + 1.0) < 1e-3 and not t["eddies"] and t["mass_defect"] < 1e-3


@pytest.mark.gpu
def test_ev_nsfnet_is_class2_and_plain_nsfnet_a_different_closed_vortex(monkeypatch, tmp_path):
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("NSFNET_PRECISION", "bf16x3")
    X, Y, U, V = ft.load_dns(os.path.join(DNS, "cavity_Re2000_256.mat"))
    dns = ft.topology(X, Y, U, V)
    tr = os.path.join(HERE, "golden", "trained")
    # ---- ev-NSFnet (6x80 + 4x40, 2.7 M steps of this engine): the DNS flow = Class 2 ----
    u, v = ft.predict_field("ev", os.path.join(tr, "ev_re2000_6x80_net.pth"), X, Y, 6, 80)
    ev = ft.topology(X, Y, u, v)
    assert abs(ev["primary"]["x"] - dns["primary"]["x"]) < 0.02 and abs(ev["primary"]["y"] - dns["primary"]["y"]) < 0.02
    assert abs(ev["primary"]["psi_min"] - dns["primary"]["psi_min"]) < 0.05 * abs(dns["primary"]["psi_min"])
    assert [e["where"] for e in ev["eddies"][:2]] == ["bottom-right", "bottom-left"]
    for a, b in zip(ev["eddies"][:2], dns["eddies"][:2]):
        assert abs(a["x"] - b["x"]) < 0.03 and abs(a["y"] - b["y"]) < 0.03, (a, b)
    assert np.linalg.norm(u - U) / np.linalg.norm(U) < 0.04
    # ---- plain NSFnet (4x120, the reference's 1.6 M-step schedule): a different steady solution ----
    from nsfnet_amd import pinn_solver as ps
    from oracle import autograd_ref as ar
    net = os.path.join(tr, "nsfnet_re2000_4x120_net.pth")
    u1, v1 = ft.predict_field("nsfnet", net, X, Y, 4, 120)
    c1 = ft.topology(X, Y, u1, v1)
    P = ps.PysicsInformedNeuralNetwork(Re=2000, layers=4, hidden_size=120, N_f=66049, bc_weight=10, eq_weight=1,
                                       net_params=net)
    P.set_boundary_data(X=ar.cavity_boundary())
    # residuals on 60 000 FRESH uniform points (not the training set) of [0.01, 0.99]^2.  The 1 % band along the walls is
    # left out: within 0.5 % of the lid's corners the continuity residual of this field reaches O(1) (max 2.0 at
    # (0.995, 0.998); mean over the whole square 2.2e-4 against 3e-7 without the band) - the regularised lid still
    # meets the side wall in a corner the 4x120 net does not resolve; measured on MI355X, round 2.
    rng = np.random.RandomState(2024)
    P.set_eq_training_data(X=(0.01 + 0.98 * rng.rand(60000, 1), 0.01 + 0.98 * rng.rand(60000, 1)))
    loss, (loss_e, loss_b) = P.fwd_computing_loss_2d()
    res = [float(P.loss_eq1), float(P.loss_eq2), float(P.loss_eq3)]
    assert max(res) < 1e-5, res                                                                    # a Navier-Stokes solution
    assert float(loss_b) < 1e-6 and c1["mass_defect"] < 2e-3                                       # of the same BVP
    assert c1["primary"]["psi_min"] < -0.05                                                        # one strong closed vortex
    assert np.hypot(c1["primary"]["x"] - dns["primary"]["x"], c1["primary"]["y"] - dns["primary"]["y"]) > 0.1   # elsewhere
    assert not any(e["where"].startswith("bottom") and e["area"] < 0.1 for e in c1["eddies"])     # no DNS corner eddies
    assert max(e["area"] for e in c1["eddies"]) > 0.25                   # instead: one large counter-rotating region
    assert np.linalg.norm(u1 - U) / np.linalg.norm(U) > 0.5
    rec = json.load(open(os.path.join(ROOT, "profiles", "r02_convergence_nsfnet_re2000_summary.json")))
    assert abs(rec["topology"]["primary"]["x"] - c1["primary"]["x"]) < 5e-3       # the committed record is this field
