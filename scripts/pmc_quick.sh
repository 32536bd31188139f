#!/bin/bash
# one SQ counter pass over the default bench with a given library:  scripts/pmc_quick.sh <tag> [lib.so]
TAG=$1; LIB=${2:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
[ -n "$LIB" ] && export NSFNET_PINN_LIB=$ROOT/$LIB
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq1 -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --alt-precision "" --sustain-seconds 0 > $OUT/sq1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --alt-precision "" --sustain-seconds 0 > $OUT/sq2.log 2>&1
python3 $ROOT/scripts/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
grep -A18 "^fwd_pipe_kernel<256, 3>" $OUT/summary.txt
