#!/bin/bash
# sample socket power / clocks while the default bench loops:  bash scripts/power_sample.sh [bench args]   (GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/power_sample.txt
rocm-smi --showmaxpower --showpower --showclocks > $OUT 2>&1
python3 $ROOT/bench.py --steps 1500 --warmup 20 --no-cpu-baseline --alt-precision "" --sustain-seconds 0 "$@" > $ROOT/gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 45
for i in 1 2 3 4 5 6 7 8; do
  echo "--- sample $i" >> $OUT
  rocm-smi --showpower --showclocks --showuse 2>&1 | grep -E "Power|sclk|mclk|fclk|busy" >> $OUT
  sleep 1
done
wait $BP
tail -1 $ROOT/gpurun_out/power_bench.log >> $OUT
