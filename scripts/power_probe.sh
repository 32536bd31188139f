#!/bin/bash
# Board power and shader clock (rocm-smi / amd-smi, once a second) while forward + reverse sweep + dW loop.
#   bash scripts/power_probe.sh > gpurun_out/power_probe.txt        POWER_CASES="precision,layers,hidden,grid,zero ..." (zero=1: all-zero parameters)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" || true
sample() {
  for i in 1 2 3 4; do
    rocm-smi --showpower 2>/dev/null | grep -i "power (W)" | sed "s/^/  $1 rocm-smi /"
    rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk" | sed "s/^/  $1 rocm-smi /"
    (amd-smi metric -p 2>/dev/null | grep -i -E "SOCKET_POWER" | sed "s/^/  $1 amd-smi /") || true
    sleep 1
  done
}
# cases: "precision layers hidden grid zero"
CASES=${POWER_CASES:-"bf16x3,6,256,600,0 bf16x3,6,256,600,1"}
for case in $CASES; do
  IFS=, read prec nl nh ng zero <<< "$case"
  z=""; [ "$zero" = 1 ] && z="--zero"
  tag="${prec}-${nl}x${nh}-${ng}${z}"
  rm -f /tmp/power_probe_go
  python3 - <<PY &
import os, sys, time
sys.argv = ["abl_time.py", "--what", "step"] + ("$z".split() if "$z" else [])
sys.path.insert(0, "$ROOT/scripts"); sys.path.insert(0, "$ROOT")
import numpy as np, torch, bench
from nsfnet_amd import engine as eng
dev = torch.device("cuda:0")
e = eng.PinnEngine(dev, $nl, $nh, 2000.0, alpha_b=10.0, alpha_e=1.0, precision="$prec")
e.net.set_flat(bench.seeded_flat($nl, $nh) * (0.0 if "$z" else 1.0))
x, y = bench.grid_block($ng, $ng, 0, 1); xb, yb, ub, vb = bench.cavity_boundary()
e.set_collocation(x, y); e.set_boundary(xb, yb, ub, vb)
f = e.plan_f; c = 2.0 / x.size
open('/tmp/power_probe_go', 'w').close()
t0 = time.time(); n = 0
while time.time() - t0 < 15:
    for _ in range(5):
        f.forward(2000.0, save=True); f.backward(2000.0, (c, c, c, 0.0))
    torch.cuda.synchronize(); n += 5
print("loop ${tag}: %.3f ms per forward + reverse sweep + dW" % (1e3 * (time.time() - t0) / n), flush=True)
PY
  for i in $(seq 1 150); do [ -e /tmp/power_probe_go ] && break; sleep 1; done; sleep 3
  sample "${tag}"
  wait
done
