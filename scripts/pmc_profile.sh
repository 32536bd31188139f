#!/bin/bash
# PMC passes over the default bench (run on the GPU box):  scripts/pmc_profile.sh <precision> <tag> [bench args]
# One rocprofv3 run per counter group (SQ has 8 slots, FETCH_SIZE/WRITE_SIZE need separate TCC passes);
# --pmc is never combined with tracing modes.  Output: gpurun_out/pmc_<tag>/<group>/...counter_collection.csv,
# gpurun_out/pmc_<tag>/summary.txt and gpurun_out/pmc_<tag>/pmc.json (copy the latter two into profiles/).
set -u
PREC=${1:-bf16x3}; TAG=${2:-r2}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --alt-precision "" --sustain-seconds 0 --precision $PREC $EXTRA > $OUT/$name.log 2>&1
  echo "pass $name rc=$?" >> $OUT/passes.log
}
EXTRA="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
cp -f $ROOT/profiles/r03_pmc.json $OUT/pmc.json 2>/dev/null
# (the shape label follows the bench arguments: `--config N` in EXTRA selects that BASELINE shape)
CFG=$(echo " $EXTRA" | sed -n 's/.*--config[ =]\([0-9]\).*/--config \1/p')
python3 $ROOT/scripts/pmc_summarize.py $OUT --json $OUT/pmc.json --precision $PREC $CFG > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
