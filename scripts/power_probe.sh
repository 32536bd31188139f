#!/bin/bash
# Is board power readable on the box?  Samples every source it can find while the three headline kernels loop on the
# seeded weights and on all-zero parameters (scripts/abl_time.py).   bash scripts/power_probe.sh > gpurun_out/power_probe.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ls /sys/class/drm/ 2>/dev/null | head -20
for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_input /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap; do [ -r "$f" ] && echo "$f $(cat $f 2>/dev/null)"; done
sample() {
  for i in 1 2 3 4 5 6; do
    for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_input; do [ -r "$f" ] && echo "  $1 $(basename $(dirname $(dirname $(dirname $f)))) $(cat $f 2>/dev/null) uW"; done
    rocm-smi --showpower 2>/dev/null | grep -i "power (W)" | sed "s/^/  $1 rocm-smi /"
    (amd-smi metric -p 2>/dev/null | grep -i -E "SOCKET_POWER|power" | head -3 | sed "s/^/  $1 amd-smi /") || true
    sleep 1
  done
}
for z in "" "--zero"; do
  python3 - <<PY &
import os, sys, time
sys.argv = ["abl_time.py", "--what", "step"] + ("$z".split() if "$z" else [])
sys.path.insert(0, "$ROOT/scripts"); sys.path.insert(0, "$ROOT")
import numpy as np, torch, bench
from nsfnet_amd import engine as eng
dev = torch.device("cuda:0")
e = eng.PinnEngine(dev, 6, 256, 2000.0, alpha_b=10.0, alpha_e=1.0, precision="bf16x3")
e.net.set_flat(bench.seeded_flat(6, 256) * (0.0 if "$z" else 1.0))
x, y = bench.grid_block(600, 600, 0, 1); xb, yb, ub, vb = bench.cavity_boundary()
e.set_collocation(x, y); e.set_boundary(xb, yb, ub, vb)
f = e.plan_f; c = 2.0 / x.size
t0 = time.time(); n = 0
while time.time() - t0 < 14:
    for _ in range(20):
        f.forward(2000.0, save=True); f.backward(2000.0, (c, c, c, 0.0))
    torch.cuda.synchronize(); n += 20
print("loop${z}: %.3f ms per forward + reverse sweep + dW" % (1e3 * (time.time() - t0) / n), flush=True)
PY
  sleep 6
  sample "kernels${z}"
  wait
done
