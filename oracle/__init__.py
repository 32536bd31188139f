"""CPU oracle for the PINN training-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and there only as the
checker / reported CPU baseline - never as the thing measured or shipped.
``nsfnet_amd`` never imports this package: its hot path is the HIP library
and fails loudly when that library is missing.

Parity pinning: the reference (latteine1217/NSFnet) ships no tests, golden
vectors or known-answer fixtures for this path, so the
oracle is pinned against outputs of the reference itself, imported on CPU in
the build container by ``oracle/gen_golden.py`` and committed as small
``tests/golden/*.npz`` fixtures.  ``tests/test_oracle_golden.py`` checks both
restatements here against those fixtures.

Modules
  autograd_ref  torch restatement of the reference algorithm (9 reverse-mode
                ``autograd.grad`` sweeps + ``backward`` + Adam); this is the
                "reference CPU path" timed by bench.py's cpu_baseline.
  fwdmode_ref   numpy fp64 forward-mode (value, d/dx, d/dy, Laplacian) sweep
                with a hand-derived reverse pass - the executable spec of the
                HIP kernels.
"""
