# usage: bash scripts/run_abl.sh WHAT variant...   (libraries from scripts/abl_build.py; times into gpurun_out/abl_WHAT.txt)
what=$1; shift
for v in "$@"; do NSFNET_PINN_LIB=experiments/abl/lib_$v.so timeout -k 10 120 python scripts/abl_time.py --what $what --tag $v >> gpurun_out/abl_$what.txt 2>&1 || exit 1; done
