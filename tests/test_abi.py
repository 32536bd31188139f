"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/nsfnet_pinn.h declares; host-only entry points validate their arguments;
the Python side fails loudly when the library is missing (no fallback)."""
import ctypes
import os
import re

import pytest

from oracle import fwdmode_ref as fr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from nsfnet_amd import build, _lib
    build.build()
    return _lib.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "nsfnet_pinn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pinn_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from nsfnet_amd import _lib
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names


def test_host_only_entry_points(lib):
    assert lib.pinn_abi_version() == 3
    h = ctypes.c_void_p()
    assert lib.pinn_net_create(3, 6, 256, ctypes.byref(h)) == 0
    assert lib.pinn_net_num_params(h) == fr.param_count(2, 3, 6, 256) == 330499
    assert lib.pinn_net_prep_floats(h) > 2 * 5 * 256 * 256
    p = ctypes.c_void_p()
    assert lib.pinn_plan_create(h, 360000, 4, ctypes.byref(p)) == 0
    assert lib.pinn_plan_padded_points(p) == 360000
    assert lib.pinn_plan_workspace_bytes(p, 1) > lib.pinn_plan_workspace_bytes(p, 0) > 0
    q = ctypes.c_void_p()
    assert lib.pinn_plan_create(h, 2052, 1, ctypes.byref(q)) == 0
    assert lib.pinn_plan_padded_points(q) == 2112          # fp32 @ hidden 256: 33 tiles of 64 points
    assert lib.pinn_net_set_precision(h, 1, 1, 1) == 0     # bf16x3: 128-column tiles
    q2 = ctypes.c_void_p()
    assert lib.pinn_plan_create(h, 2052, 1, ctypes.byref(q2)) == 0
    assert lib.pinn_plan_padded_points(q2) == 2176         # 17 tiles of 128 points
    p2 = ctypes.c_void_p()
    assert lib.pinn_plan_create(h, 360000, 4, ctypes.byref(p2)) == 0
    assert [lib.pinn_plan_kernel(p2, k) for k in (0, 1, 2)] == [b"fwd_split_kernel", b"bwd_split_kernel", b"dw_bf16_kernel"]
    assert [lib.pinn_plan_kernel(p, k) for k in (0, 1, 2)] == [b"fwd_wide_kernel", b"bwd_wide_kernel", b"dw_wide_kernel"]
    assert lib.pinn_plan_kernel(q2, 0) == b"fwd_bf16_kernel" and lib.pinn_plan_kernel(p2, 3) is None
    lib.pinn_plan_destroy(p2)
    # the role-split sweeps share a spill format without the layer-0 slot: they come as a pair (with the bf16 dW kernel);
    # asked for on one sweep only, that sweep runs the pipelined schedule
    import os
    for env, want in (({"PINN_FWD_SCHED": "2", "PINN_BWD_SCHED": "0"}, [b"fwd_pipe_kernel", b"bwd_bf16_kernel"]),
                      ({"PINN_FWD_SCHED": "1", "PINN_BWD_SCHED": "2"}, [b"fwd_pipe_kernel", b"bwd_pipe_kernel"]),
                      ({"PINN_SCHED": "0"}, [b"fwd_bf16_kernel", b"bwd_bf16_kernel"])):
        old = {k: os.environ.get(k) for k in ("PINN_SCHED", "PINN_FWD_SCHED", "PINN_BWD_SCHED")}
        try:
            for k in old:
                os.environ.pop(k, None)
            os.environ.update(env)
            p3 = ctypes.c_void_p()
            assert lib.pinn_plan_create(h, 360000, 4, ctypes.byref(p3)) == 0
            assert [lib.pinn_plan_kernel(p3, k) for k in (0, 1)] == want, env
            lib.pinn_plan_destroy(p3)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
    lib.pinn_plan_destroy(q2)
    for handle in (p, q):
        assert lib.pinn_plan_destroy(handle) == 0
    assert lib.pinn_net_destroy(h) == 0
    for n_out, L, H in ((3, 4, 50), (1, 4, 40), (3, 1, 7)):
        assert lib.pinn_net_create(n_out, L, H, ctypes.byref(h)) == 0
        assert lib.pinn_net_num_params(h) == fr.param_count(2, n_out, L, H)
        lib.pinn_net_destroy(h)


def test_workspace_is_sized_for_what_the_plan_writes(lib, monkeypatch):
    """The role-split pair spills three 16-byte planes per register quad and no layer 0: its S and Z-bar are (L - 1) blocks of
    3/4 of the classic HP x 128 floats per tile.  Every other plan keeps the classic [tile][L] blocks.  Wide nets: the
    kernel choice (role-split at 64-column tiles for hidden 288..448, 8-wave above) does not change the workspace."""
    for k in ("PINN_SCHED", "PINN_FWD_SCHED", "PINN_BWD_SCHED", "PINN_WSPLIT"):
        monkeypatch.delenv(k, raising=False)
    h = ctypes.c_void_p()
    assert lib.pinn_net_create(3, 6, 256, ctypes.byref(h)) == 0
    assert lib.pinn_net_set_precision(h, 1, 1, 1) == 0
    p = ctypes.c_void_p()
    assert lib.pinn_plan_create(h, 360000, 4, ctypes.byref(p)) == 0
    ntiles, L, HP = 360000 // 32, 6, 256
    spill = 2 * (ntiles + 1) * (L - 1) * 3 * HP * 32 * 4               # S + Z-bar
    total = lib.pinn_plan_workspace_bytes(p, 1)
    assert spill <= total <= spill + 200 * 1024 * 1024, (spill, total)     # + dW slabs, skinny accumulators, partials
    assert total < 11.4e9
    lib.pinn_plan_destroy(p)
    monkeypatch.setenv("PINN_SCHED", "0")                                   # classic layout: 4 planes x L layers
    assert lib.pinn_plan_create(h, 360000, 4, ctypes.byref(p)) == 0
    classic = lib.pinn_plan_workspace_bytes(p, 1)
    assert classic >= 2 * (ntiles + 1) * L * HP * 128 * 4 > 1.55 * spill
    lib.pinn_plan_destroy(p); lib.pinn_net_destroy(h)
    monkeypatch.delenv("PINN_SCHED")
    sizes = {}
    for ws in ("1", "0"):
        monkeypatch.setenv("PINN_WSPLIT", ws)
        assert lib.pinn_net_create(3, 8, 400, ctypes.byref(h)) == 0
        assert lib.pinn_net_set_precision(h, 1, 1, 1) == 0
        assert lib.pinn_plan_create(h, 500000, 4, ctypes.byref(p)) == 0
        names = [lib.pinn_plan_kernel(p, k) for k in (0, 1, 2)]
        assert names == ([b"fwd_wsplit_kernel", b"bwd_wsplit_kernel", b"dw_bf16_wide_kernel"] if ws == "1" else
                         [b"fwd_bf16_wide_kernel", b"bwd_bf16_wide_kernel", b"dw_bf16_wide_kernel"])
        sizes[ws] = lib.pinn_plan_workspace_bytes(p, 1)
        lib.pinn_plan_destroy(p); lib.pinn_net_destroy(h)
    # (the pair grid of the role-split sweeps changes the skinny-accumulator block by a few MB, nothing else)
    assert abs(sizes["1"] - sizes["0"]) < 64 * 1024 * 1024
    for H, want in ((480, b"fwd_bf16_wide_kernel"), (448, b"fwd_wsplit_kernel"), (288, b"fwd_wsplit_kernel")):
        monkeypatch.setenv("PINN_WSPLIT", "1")
        assert lib.pinn_net_create(3, 3, H, ctypes.byref(h)) == 0
        assert lib.pinn_net_set_precision(h, 1, 1, 1) == 0
        assert lib.pinn_plan_create(h, 1000, 4, ctypes.byref(p)) == 0
        assert lib.pinn_plan_kernel(p, 0) == want, H
        lib.pinn_plan_destroy(p); lib.pinn_net_destroy(h)


def test_argument_errors_are_reported(lib):
    h = ctypes.c_void_p()
    assert lib.pinn_net_create(3, 6, 600, ctypes.byref(h)) < 0
    assert b"hidden width" in lib.pinn_last_error()
    assert lib.pinn_net_create(3, 8, 400, ctypes.byref(h)) == 0          # wide nets: bf16 sweeps, fp32 dW
    assert lib.pinn_net_set_precision(h, 1, 1, 1) == 0
    assert lib.pinn_net_set_precision(h, 3, 0, 0) < 0 and b"precision must be" in lib.pinn_last_error()
    assert lib.pinn_net_set_precision(h, 0, 0, 0) == 0
    lib.pinn_net_destroy(h)
    assert lib.pinn_net_create(4, 6, 64, ctypes.byref(h)) < 0
    assert lib.pinn_net_create(3, 0, 64, ctypes.byref(h)) < 0
    assert lib.pinn_net_create(3, 2, 16, ctypes.byref(h)) == 0
    p = ctypes.c_void_p()
    assert lib.pinn_plan_create(h, 100, 2, ctypes.byref(p)) < 0
    assert lib.pinn_plan_create(h, 0, 4, ctypes.byref(p)) < 0
    assert lib.pinn_plan_create(h, 100, 1, ctypes.byref(p)) == 0
    # a value plan is refused by the residual entry point before anything is launched
    rc = lib.pinn_residual_forward(p, 1, 1, 1, 1, None, None, None, None, 1, 100.0, 0.0, 0.0, 1.0, 0, None, None)
    assert rc < 0 and b"not a residual" in lib.pinn_last_error()
    assert lib.pinn_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 1, None) < 0
    lib.pinn_plan_destroy(p); lib.pinn_net_destroy(h)


def test_no_cached_state_in_the_library_sources():
    """include/nsfnet_pinn.h promises no global mutable state besides the (thread-local) last-error
    string: no function-local or file-scope `static` variables in csrc/ (launch attributes are set per
    plan, on the plan's device, by pinn_plan_create)."""
    csrc = os.path.join(ROOT, "nsfnet_amd", "csrc")
    bad = []
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        for i, line in enumerate(open(os.path.join(csrc, f)), 1):
            m = re.match(r"\s*static\s+(?!constexpr|inline|thread_local|__device__|__global__)([\w:<> ]+?)\s+(\w+)\s*(=|;|\[)", line)
            if m:
                bad.append("%s:%d: %s" % (f, i, line.strip()))
    assert not bad, bad
    text = open(os.path.join(csrc, "capi.hip")).read()
    assert len(re.findall(r"thread_local", text)) == 1          # the last-error buffer


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nsfnet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PinnLibraryError, match="no CPU fallback"):
        _lib.load()


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "nsfnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
