# usage: bash scripts/run_abl_wide.sh WHAT variant...   (config 5's shape: 8x400, 707x707 points; times into gpurun_out/abl_wide_WHAT.txt)
what=$1; shift
for v in "$@"; do NSFNET_PINN_LIB=experiments/abl/lib_$v.so timeout -k 10 150 python scripts/abl_time.py --layers 8 --hidden 400 --grid 707 --what $what --tag $v >> gpurun_out/abl_wide_$what.txt 2>&1 || exit 1; done
