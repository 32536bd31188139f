"""Chunked collocation passes (nsfnet_amd.engine.ChunkedResidual: SURVEY 8(f4), the reference's
mini-batching roadmap item) keep FULL-BATCH semantics: same loss terms, gradients, lagged viscosity
and parameters as the everything-resident path, with one shared activation workspace.
CPU: host logic on the oracle-backed fakes.  GPU: the real kernels."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ev_solver(monkeypatch, chunk, n=300):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fakes
    fakes.install(monkeypatch)
    from nsfnet_amd import ev_pinn_solver as es
    from oracle import autograd_ref as ar
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    if chunk:
        monkeypatch.setenv("NSFNET_CHUNK_POINTS", str(chunk))
    else:
        monkeypatch.delenv("NSFNET_CHUNK_POINTS", raising=False)
    rng = np.random.RandomState(7)
    x, y = rng.rand(n, 1), rng.rand(n, 1)
    w = (0.5 + rng.rand(n)).astype(np.float32)
    xb, yb, ub, vb = (a[::63][:33] for a in ar.cavity_boundary())
    torch.manual_seed(3)
    P = es.PysicsInformedNeuralNetwork(Re=800, layers=2, layers_1=2, hidden_size=10, hidden_size_1=6, N_f=n,
                                       alpha_evm=0.05, bc_weight=10, eq_weight=1)
    P.set_boundary_data(X=(xb, yb, ub, vb))
    P.set_eq_training_data(X=(x, y), weights=w)
    P.log_interval = 1000
    P.save = lambda *a, **k: None
    return P


def test_chunked_host_logic_matches_single_pass(monkeypatch):
    from nsfnet_amd import engine as eng
    out = []
    for chunk in (0, 128):
        P = _ev_solver(monkeypatch, chunk)
        assert isinstance(P.engine.plan_f, eng.ChunkedResidual) == bool(chunk)
        if chunk:
            assert [b - a for a, b in P.engine.plan_f.bounds] == [128, 128, 44]
        P.freeze_evm_net(0)
        rec = []
        for k in range(3):
            P.engine.e_trainable = (k == 1)          # one step with the entropy net in the gradient
            loss, _ = P.fwd_computing_loss_2d()
            rec.append([float(loss), float(P.loss_e), float(P.loss_b), float(P.loss_eq4)])
            P.engine.adam_step(1e-3)
        out.append(dict(rec=np.array(rec), p=P.engine.net.params.numpy().copy(), pe=P.engine.net_e.params.numpy().copy(),
                        vt=P.engine.plan_f.vis_t.numpy().copy(), vtm=P.engine.plan_f.vis_t_minus.numpy().copy(),
                        eq1=P.eq1_pred.numpy().copy()))
    a, b = out
    np.testing.assert_allclose(b["rec"], a["rec"], rtol=2e-6)
    np.testing.assert_allclose(b["p"], a["p"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(b["pe"], a["pe"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(b["vt"], a["vt"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(b["vtm"], a["vtm"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(b["eq1"], a["eq1"], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_chunked_kernels_match_single_pass(prec):
    from nsfnet_amd import engine as eng
    from oracle import autograd_ref as ar
    dev = torch.device("cuda:0")
    L, H, N = 3, 64, 5000
    rng = np.random.RandomState(1)
    x, y = rng.rand(N).astype(np.float32), rng.rand(N).astype(np.float32)
    w = (0.5 + rng.rand(N)).astype(np.float32)
    xb, yb, ub, vb = (a.reshape(-1)[::8].astype(np.float32) for a in ar.cavity_boundary())
    flat = ar.flat_params(ar.seeded_net(3, L, H, seed=11)).numpy().copy()
    flat_e = ar.flat_params(ar.seeded_net(1, 2, 24, seed=12)).numpy().copy()
    res = []
    for chunk in (None, 1024):
        E = eng.PinnEngine(dev, L, H, 3000.0, alpha_b=10.0, alpha_e=1.0, flavour="ev", n_hidden_e=2, hidden_e=24,
                           alpha_evm=0.05, precision=prec)
        E.net.set_flat(torch.tensor(flat)); E.net_e.set_flat(torch.tensor(flat_e))
        E.set_collocation(x, y, weights=w, chunk_points=chunk)
        E.set_boundary(xb, yb, ub, vb)
        assert isinstance(E.plan_f, eng.ChunkedResidual) == (chunk is not None)
        if chunk:      # one workspace for all five passes
            assert len({c.ws.data_ptr() for c in E.plan_f.chunks}) == 1 and len(E.plan_f.chunks) == 5
        E.e_trainable = True
        for _ in range(2):        # second step exercises the lagged viscosity state per chunk
            E.loss_and_grad()
            terms = {k: float(v) for k, v in E.loss_terms().items()}
            g, ge = E.grads.cpu().numpy().copy(), E.grads_e.cpu().numpy().copy()
            E.adam_step(1e-3)
        torch.cuda.synchronize()
        res.append(dict(terms=terms, g=g, ge=ge, eq2=E.plan_f.field("eq2").cpu().numpy(),
                        vt=E.plan_f.vis_t.cpu().numpy(), p=E.net.params.cpu().numpy()))
    a, b = res
    for k in a["terms"]:
        assert abs(a["terms"][k] - b["terms"][k]) <= 2e-6 * abs(a["terms"][k]) + 1e-12, k
    # tile boundaries move with the chunking, so sums re-associate: fp32 round-off only
    assert np.linalg.norm(a["g"] - b["g"]) <= 2e-6 * np.linalg.norm(a["g"])
    assert np.linalg.norm(a["ge"] - b["ge"]) <= 2e-5 * np.linalg.norm(a["ge"])
    np.testing.assert_allclose(b["eq2"], a["eq2"], rtol=0, atol=1e-5 * np.abs(a["eq2"]).max())
    np.testing.assert_allclose(b["vt"], a["vt"], rtol=1e-6)
    np.testing.assert_allclose(b["p"], a["p"], rtol=0, atol=1e-6)


def test_precision_spec_parsing(monkeypatch):
    from nsfnet_amd import engine as eng
    monkeypatch.delenv("NSFNET_PRECISION", raising=False)
    assert eng.resolve_precision() == ("fp32", "fp32", "fp32")
    assert eng.resolve_precision("bf16x3") == ("bf16x3",) * 3
    assert eng.resolve_precision("bf16x3,fp32,bf16") == ("bf16x3", "fp32", "bf16")
    monkeypatch.setenv("NSFNET_PRECISION", "bf16")
    assert eng.resolve_precision() == ("bf16",) * 3
    with pytest.raises(ValueError):
        eng.resolve_precision("fp16")
    with pytest.raises(ValueError):
        eng.resolve_precision("fp32,bf16")


def test_chunk_bounds_are_tile_aligned(monkeypatch):
    """Chunk boundaries sit on multiples of 128 points (every tile size divides it), chunks cover the set once,
    and a chunk size below the alignment is rounded up to it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fakes
    fakes.install(monkeypatch)
    from nsfnet_amd import engine as eng
    net = eng.DeviceNet(3, 2, 8, torch.device("cpu"))
    x = np.linspace(0, 1, 1000); y = x[::-1].copy()
    for chunk, sizes in ((300, [256, 256, 256, 232]), (50, [128] * 7 + [104]), (4096, [1000])):
        c = eng.ChunkedResidual(net, x, y, None, chunk)
        assert [b - a for a, b in c.bounds] == sizes
        assert c.bounds[0][0] == 0 and c.bounds[-1][1] == 1000
        assert all(a % 128 == 0 for a, _ in c.bounds)
        assert c.n == 1000 and len({id(p.ws) for p in c.chunks}) == 1
