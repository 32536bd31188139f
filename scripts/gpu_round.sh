#!/bin/bash
# One GPU-box session of the round's routine evidence:  bash scripts/gpu_round.sh TAG [steps ...]
#   tests   pytest -m gpu                                  -> gpurun_out/TAG_gputest.log
#   bench   default bench.py                               -> gpurun_out/TAG_bench.json
#   stats   rocprofv3 --kernel-trace --stats of the bench  -> gpurun_out/TAG_kernel_stats.csv
#   pmc     scripts/pmc_profile.sh (separate --pmc passes) -> gpurun_out/pmc_TAG/{summary.txt,pmc.json}
#   clock   GRBM_GUI_ACTIVE of the three kernels on seeded and on all-zero parameters -> gpurun_out/TAG_clock.txt
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
for step in "$@"; do
  cd $ROOT
  case $step in
    tests) timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputest.log 2>&1 || { tail -30 gpurun_out/${TAG}_gputest.log; exit 1; }; tail -2 gpurun_out/${TAG}_gputest.log ;;
    bench) python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench.json ;;
    stats) cd /tmp && export TMPDIR=/tmp
           rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --alt-precision "" --sustain-seconds 0 > $ROOT/gpurun_out/${TAG}_stats.log 2>&1
           cp $(ls $ROOT/gpurun_out/${TAG}_stats/*/*kernel_stats.csv | head -1) $ROOT/gpurun_out/${TAG}_kernel_stats.csv; head -8 $ROOT/gpurun_out/${TAG}_kernel_stats.csv ;;
    pmc)   bash scripts/pmc_profile.sh bf16x3 $TAG ;;
    pmc5)  cp -f gpurun_out/pmc_$TAG/pmc.json profiles/r03_pmc.json 2>/dev/null || true      # (config 5's entries join the same record)
           bash scripts/pmc_profile.sh bf16x3 ${TAG}c5 --config 5 ;;
    configs) for c in 1 2 4 5; do python bench.py --config $c > gpurun_out/${TAG}_bench_config$c.json 2> gpurun_out/${TAG}_bench_config$c.err; python -c "
import json,sys; d=json.load(open('gpurun_out/${TAG}_bench_config$c.json')); print('config $c', round(d['ms_per_step'],3), 'ms', d['roofline']['kernel_ms'], round(d['roofline']['frac'],4))"; done ;;
    fuzz)  timeout -k 10 600 python scripts/fuzz_shapes.py 3 40 > gpurun_out/${TAG}_fuzz.txt 2>&1; tail -3 gpurun_out/${TAG}_fuzz.txt ;;
    strong) python bench.py --scaling strong --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/${TAG}_bench_strong1.json 2> gpurun_out/${TAG}_bench_strong1.err; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_strong1.json')); print('strong N=1', d['config']['global_points'], round(d['ms_per_step'],3), 'ms', d['value'])" ;;
    big)   python bench.py --grid 2000 --steps 3 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --alt-precision "" > gpurun_out/${TAG}_bench_grid2000.json 2> gpurun_out/${TAG}_bench_grid2000.err; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_grid2000.json')); print('4 M points', round(d['ms_per_step'],2), 'ms', d['value'])"; grep -i "workspace\|GiB\|memory" gpurun_out/${TAG}_bench_grid2000.err | tail -3 ;;
    clock) cd /tmp && export TMPDIR=/tmp
           for z in "" "--zero"; do
             rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_clock$z -- python3 $ROOT/scripts/abl_time.py --what fwd,bwd,dw --tag "seeded$z" $z >> $ROOT/gpurun_out/${TAG}_clock.txt 2>&1
           done
           python3 $ROOT/scripts/clock_from_pmc.py $ROOT/gpurun_out/${TAG}_clock $ROOT/gpurun_out/${TAG}_clock--zero >> $ROOT/gpurun_out/${TAG}_clock.txt; grep -v amdgpu.ids $ROOT/gpurun_out/${TAG}_clock.txt ;;
  esac
done
