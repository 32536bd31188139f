#!/bin/bash
# Board power and shader clock per MFMA operand type: tests/micro/mfma_dtype_power loops each type for SECONDS while this
# script samples rocm-smi and tags every sample with the type running at that moment.
#   bash scripts/mfma_power.sh [SECONDS] > gpurun_out/mfma_power.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
SECS=${1:-4}
LOG=$(mktemp)
"$ROOT/tests/micro/mfma_dtype_power" "$SECS" > "$LOG" 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do
  tag=$(grep "^BEGIN" "$LOG" | tail -1 | sed 's/^BEGIN //')
  n_begin=$(grep -c "^BEGIN" "$LOG"); n_end=$(grep -c "^END" "$LOG")
  p=$(rocm-smi --showpower 2>/dev/null | grep -i "power (W)" | sed 's/.*: //')
  c=$(rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | sed 's/.*(\(.*\))/\1/')
  [ "$n_begin" -gt "$n_end" ] && echo "  sample [$tag] $p W  sclk $c"
done
wait $PID
grep "^END" "$LOG"
rm -f "$LOG"
