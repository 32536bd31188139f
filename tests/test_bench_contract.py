"""bench.py keeps the driver's contract: exactly one JSON line on stdout with the required keys, the roofline and
(at N = 1) cpu_baseline objects, measured at a reduced size so the test stays short; `python bench.py --gpus N`
starts its own ranks (as children, through torch.distributed.run on 127.0.0.1) when no launcher did."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "sustained")


def _one_json_line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_common(d, n_gpus, steps, warmup):
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_points"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    s = d["sustained"]
    assert s["seconds"] > 0 and s["steps"] > 0 and abs(s["ms_per_step"] - 1e3 * s["seconds"] / s["steps"]) < 1e-9


@pytest.mark.gpu
def test_bench_json_line():
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--grid", "128", "--cpu-sample", "1024", "--sustain-seconds", "0.2"],
                         capture_output=True, text=True, env=env, timeout=600)
    d = _one_json_line(out)
    _check_common(d, 1, 3, 1)
    assert d["config"]["workload"].startswith("custom")        # not a BASELINE shape: says so
    assert d["config"]["backend"] == "none" and "alts" not in d
    assert d["roofline"]["traffic"] is None                     # PMC figures only for the shape they were measured on
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["4", "5"])
def test_bench_ev_configs_run_under_the_json_contract(config):
    """BASELINE configs 4 and 5 (ev-NSFnet, ev-NSFnet/pinn_solver.py:290-342) through the driver-visible bench,
    at a reduced per-GPU grid."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--grid", "64x96",
                          "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--sustain-seconds", "0.1"],
                         capture_output=True, text=True, env=env, timeout=600)
    d = _one_json_line(out)
    _check_common(d, 1, 3, 1)
    assert "ev-NSFnet" in d["config"]["workload"] and "4x40 entropy net" in d["config"]["workload"]
    assert ("8x400" if config == "5" else "6x256") in d["config"]["workload"]
    assert d["config"]["final_loss"] == d["config"]["final_loss"] and d["cpu_baseline"] is None


@pytest.mark.gpu
def test_bench_self_launches_two_ranks():
    """Launched exactly as the driver does for N > 1 when no torchrun is in front: `python bench.py --gpus 2`.
    Two ranks share this box's one GPU, so the collective runs over gloo (RCCL refuses two ranks on one device);
    the rank logic - block of the global grid per rank, BC split, ONE all-reduce per step, max-over-ranks
    timing, rank 0's single JSON line - is the code the RCCL run uses."""
    env = dict(os.environ, PYTHONPATH=ROOT, NSFNET_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "64", "--steps", "4",
                          "--warmup", "2", "--sustain-seconds", "0.1"], capture_output=True, text=True, env=env, timeout=900)
    d = _one_json_line(out)
    _check_common(d, 2, 4, 2)
    assert d["config"]["global_points"] == 2 * 64 * 64 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["backend"] == "gloo"      # a rehearsal can never be mistaken for the RCCL run ("rccl")
    assert d["cpu_baseline"] is None


def test_self_launcher_starts_children_and_forwards_one_line(tmp_path):
    """CPU check of the launcher itself: N children through torch.distributed.run on 127.0.0.1, rank 0's JSON
    line forwarded alone on stdout, other output to stderr, the children's exit code returned."""
    stub = tmp_path / "bench_stub.py"
    stub.write_text(textwrap.dedent("""
        import json, os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1"
        print("noise from rank %d" % r)
        if r == 0:
            print(json.dumps({"world": w, "argv": sys.argv[1:]}))
        sys.exit(7 if "--fail" in sys.argv and r == 1 else 0)
    """))
    driver = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import bench
        bench.__file__ = %r
        sys.argv = ["bench.py"] + sys.argv[1:]
        sys.exit(bench.self_launch(2))
    """ % (ROOT, str(stub)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    ok = subprocess.run([sys.executable, "-c", driver, "--gpus", "2", "--steps", "5"], capture_output=True, text=True,
                        env=env, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    lines = [l for l in ok.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"world": 2, "argv": ["--gpus", "2", "--steps", "5"]}
    assert "noise from rank 1" in ok.stderr
    bad = subprocess.run([sys.executable, "-c", driver, "--gpus", "2", "--fail"], capture_output=True, text=True,
                         env=env, timeout=300)
    assert bad.returncode != 0


def test_strong_scaling_blocks_tile_the_fixed_total():
    """--scaling strong: config 3's net on 2.88 M points IN TOTAL, equal row blocks per rank."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    nx, ny = bench.CONFIGS[3]["grid"]
    assert bench.local_rows(nx, "weak", 4) == nx
    ref_x = (np.arange(8 * nx) + 0.5) / (8 * nx)
    for world in (1, 2, 4, 8):
        rows = bench.local_rows(nx, "strong", world)
        assert rows * world * ny == 2_880_000
        xs = []
        for r in range(world):
            x, y = bench.grid_block(rows, ny, r, world)
            assert x.size == rows * ny == 2_880_000 // world
            xs.append(np.unique(x))
        assert np.allclose(np.concatenate(xs), ref_x.astype(np.float32))        # disjoint, ordered, complete
    with pytest.raises(SystemExit):
        bench.local_rows(nx, "strong", 7)


def test_strong_scaling_two_gloo_ranks(tmp_path):
    """The N = 2 case as the driver would start it (torch.distributed.run, 127.0.0.1), on CPU over gloo: every rank
    derives its block from RANK / WORLD_SIZE alone, the blocks have 2.88 M / N points and add up to the fixed total."""
    script = tmp_path / "strong_ranks.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import numpy as np, torch, torch.distributed as dist
        import bench
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        sys.argv = ["bench.py", "--scaling", "strong", "--gpus", str(w)]
        args = bench.parse_args()
        rows = bench.local_rows(args.nx, args.scaling, w)
        x, y = bench.grid_block(rows, args.ny, r, w)
        t = torch.tensor([float(x.size), float(x.min()), float(x.max())], dtype=torch.float64)
        out = [torch.zeros_like(t) for _ in range(w)]
        dist.all_gather(out, t)
        if r == 0:
            n = sum(int(o[0]) for o in out)
            assert n == 2_880_000 and all(int(o[0]) == n // w for o in out), out
            assert out[0][2] < out[1][1]                      # rank 0's rows end before rank 1's start
            print("STRONG_OK", n, args.alt_list)
        dist.destroy_process_group()
    """ % ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--standalone", "--local-addr", "127.0.0.1", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "STRONG_OK 2880000 []" in out.stdout            # (no alt modes in the strong table)
