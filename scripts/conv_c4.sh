#!/bin/bash
# One stage of the config-4-shape convergence run (ev-NSFnet, Re = 4000, 6x256 + 4x40, 250 000 points = config 4's
# per-GPU share, bf16x3, DNS = the reference's cavity_Re4000_384_Uniform.mat):  bash scripts/conv_c4.sh STAGE SCALE
# Resumes from experiments/conv_c4_ckpt (copied there from gpurun_out/conv_c4 after the previous call).
K=$1; SCALE=${2:-0.30}
RES=""; [ "$K" -gt 1 ] && RES="--resume experiments/conv_c4_ckpt"
[ "$K" -gt 1 ] && mkdir -p gpurun_out/conv_c4 && cp experiments/conv_c4_ckpt/stages.jsonl gpurun_out/conv_c4/ 2>/dev/null
mkdir -p gpurun_out/conv_c4
# (unbuffered and un-piped: the box kills a command that writes nothing for 7 minutes)
NSFNET_PRECISION=bf16x3 timeout -k 10 1150 python -u scripts/converge_ev.py --re 4000 --dns tests/golden/dns/cavity_Re4000_384_Uniform.mat \
  --out gpurun_out/conv_c4 --first $K --last $K --epochs-scale $SCALE --nf 250000 --layers 6 --hidden 256 $RES > gpurun_out/conv_c4/log_stage$K.txt 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 60; echo "[conv_c4 stage $K] $(tail -c 300 gpurun_out/conv_c4/log_stage$K.txt | tr '\n' ' ' | tail -c 200)"; done
wait $PID; RC=$?
grep -E "STAGE|Error u|Error v" gpurun_out/conv_c4/log_stage$K.txt | tail -4
exit $RC
