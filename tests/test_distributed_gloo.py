"""world_size-2 `gloo` tests (CPU) of the multi-GPU path's HOST logic: contiguous point
shards per rank (ev-NSFnet/pinn_solver.py:142-184), ONE all-reduce of [grads | loss sums]
per step, global-count normalisation, identical replicas after Adam.  The device objects
are replaced by tests/fakes.py (oracle arithmetic on CPU); the real kernels are covered by
the -m gpu tests."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(sup_n=9):
    rng = np.random.RandomState(42)
    N, Nb = 70, 33            # neither divisible by 2: last rank takes the remainder
    x, y = rng.rand(N, 1), rng.rand(N, 1)
    from oracle import autograd_ref as ar
    xb, yb, ub, vb = (a[::63][:Nb] for a in ar.cavity_boundary())
    w = (0.5 + rng.rand(N)).astype(np.float32)
    xs, ys = rng.rand(9, 1), rng.rand(9, 1)
    us, vs, ps_ = rng.rand(9, 1), rng.rand(9, 1), rng.rand(9, 1)
    ps_[[1, 6], 0] = np.nan
    xs, ys, us, vs, ps_ = (a[:sup_n] for a in (xs, ys, us, vs, ps_))
    return dict(x=x, y=y, xb=xb, yb=yb, ub=ub, vb=vb, w=w, sup=(xs, ys, us, vs, ps_))


def _build_solver(case):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fakes
    fakes.install()
    from nsfnet_amd import ev_pinn_solver as es
    torch.manual_seed(3)
    P = es.PysicsInformedNeuralNetwork(Re=800, layers=2, layers_1=2, hidden_size=10, hidden_size_1=6, N_f=70,
                                       alpha_evm=0.05, bc_weight=10, eq_weight=1, supervised_data_weight=0.5)
    P.set_boundary_data(X=(case["xb"], case["yb"], case["ub"], case["vb"]))
    P.set_eq_training_data(X=(case["x"], case["y"]), weights=case["w"])
    P.set_supervised_data(case["sup"])
    P.set_supervised_loss_weight(0.5)
    P.log_interval = 1000
    P.save = lambda *a, **k: None
    return P


def _run_rank(rank, world, out_dir, sup_n=9):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1")
    torch.set_num_threads(1)
    # file rendezvous: no port to pick (a picked-then-closed port can be taken by another job of the host)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(out_dir, "rendezvous"), rank=rank, world_size=world)
    try:
        P = _build_solver(_case(sup_n))
        assert P.is_distributed and P.engine.world_size == world
        rec = dict(n_f_local=P.x_f.shape[0], n_b_local=P.x_b.shape[0], n_s_local=P.supervision_point_count)
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            P.train(num_epoch=3, lr=1e-3)
        rec.update(params=P.engine.net.params.numpy().copy(), loss=float(P.loss), loss_b=float(P.loss_b),
                   loss_e=float(P.loss_e), loss_s=float(P.loss_s), sums=P.engine.sums.numpy().copy())
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **rec)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_training_matches_single_process(tmp_path, monkeypatch):
    world = 2
    mp.spawn(_run_rank, args=(world, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world))
    # shards: contiguous blocks, last rank takes the remainder; supervised = np.array_split
    assert (int(r0["n_f_local"]), int(r1["n_f_local"])) == (35, 35)
    assert (int(r0["n_b_local"]), int(r1["n_b_local"])) == (16, 17)
    assert (int(r0["n_s_local"]), int(r1["n_s_local"])) == (5, 4)
    # replicas are bitwise identical after the all-reduced Adam steps
    np.testing.assert_array_equal(r0["params"], r1["params"])
    np.testing.assert_array_equal(r0["sums"], r1["sums"])
    # single process, full batch, same code path
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    P = _build_solver(_case())
    assert not P.is_distributed
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        P.train(num_epoch=3, lr=1e-3)
    np.testing.assert_allclose(r0["params"], P.engine.net.params.numpy(), rtol=0, atol=2e-6)
    for key, val in (("loss", P.loss), ("loss_b", P.loss_b), ("loss_e", P.loss_e), ("loss_s", P.loss_s)):
        assert abs(float(r0[key]) - float(val)) <= 1e-5 * abs(float(val)), key


@pytest.mark.timeout(300)
def test_rank_with_empty_supervised_share(tmp_path, monkeypatch):
    """One supervised sample over two ranks: np.array_split hands rank 1 nothing
    (ev-NSFnet/pinn_solver.py:219-221, the branch is then skipped there, :400).  That rank must still
    take part in the step's all-reduce and normalise by the global counts - not raise, not hang."""
    world = 2
    mp.spawn(_run_rank, args=(world, str(tmp_path), 1), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world))
    assert (int(r0["n_s_local"]), int(r1["n_s_local"])) == (1, 0)
    np.testing.assert_array_equal(r0["params"], r1["params"])
    np.testing.assert_array_equal(r0["sums"], r1["sums"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    P = _build_solver(_case(1))
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        P.train(num_epoch=3, lr=1e-3)
    np.testing.assert_allclose(r0["params"], P.engine.net.params.numpy(), rtol=0, atol=2e-6)
    assert float(P.loss_s) > 0 and abs(float(r0["loss_s"]) - float(P.loss_s)) <= 1e-5 * float(P.loss_s)


def test_too_few_points_for_the_ranks_raises_everywhere(monkeypatch):
    """Fewer collocation / boundary points than ranks: the error depends on global counts only, so
    every rank raises it before any collective (no rank-divergent hang)."""
    import fakes
    fakes.install(monkeypatch)
    from nsfnet_amd import ev_pinn_solver as es
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    P = es.PysicsInformedNeuralNetwork(Re=800, layers=1, layers_1=1, hidden_size=4, hidden_size_1=4, N_f=3)
    P.is_distributed, P.world_size = True, 4
    for rank in range(4):
        P.rank = rank
        with pytest.raises(ValueError, match="cannot be sharded"):
            P._shard(3)
    P.rank = 3
    assert P._shard(9) == (6, 9)


def test_single_process_fake_path_matches_autograd_oracle(monkeypatch):
    """The fakes + engine host logic reproduce the torch-autograd restatement of the reference
    step (full batch), so the gloo comparison above is anchored to the oracle."""
    from oracle import autograd_ref as ar
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    case = _case()
    P = _build_solver(case)
    P.set_supervised_data(None)
    torch.manual_seed(3)
    net = ar.RefFCNet(2, 3, 2, 10); net_e = ar.RefFCNet(2, 1, 2, 6)
    np.testing.assert_array_equal(ar.flat_params(net).numpy(), P.engine.net.params.numpy())
    o = ar.EvNSFnetOracle(net, net_e, 800.0, 0.05, alpha_b=10.0, alpha_e=1.0, lr=1e-3)
    o.set_data(case["x"], case["y"], case["xb"], case["yb"], case["ub"], case["vb"], weights=case["w"])
    P.freeze_evm_net(0)
    for k in range(3):
        P._apply_freeze_schedule(k)
        loss, _ = P.fwd_computing_loss_2d()
        ref = o.step(epoch_id=k)
        assert abs(float(loss) - ref) < 2e-5 * abs(ref)
        P.engine.adam_step(1e-3)
        np.testing.assert_allclose(P.engine.net.params.numpy(), ar.flat_params(net).numpy(), rtol=0, atol=3e-6)
