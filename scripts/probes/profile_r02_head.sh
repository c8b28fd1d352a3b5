#!/bin/bash
# the first two passes of profile_r02.sh (kernel stats of the default bench + FETCH_SIZE of cfg3), for a quick refresh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
RP="rocprofv3 --output-format csv"
timeout -k 10 900 $RP --kernel-trace --stats -d $OUT/k -- python3 bench.py --steps 20 --warmup 5 > $OUT/k_bench_under_rocprof.json 2> $OUT/k.err &&
cp $(ls $OUT/k/*/*kernel_stats.csv | head -1) $OUT/k_kernel_stats.csv &&
timeout -k 10 600 $RP --pmc FETCH_SIZE -d $OUT/f3 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-probe --no-host-probe > $OUT/f3_bench.json 2> $OUT/f3.err &&
python3 scripts/pmc_summary.py $OUT/f3 scan_ > $OUT/f3_fetch_size_cfg3.txt &&
rm -rf $OUT/k $OUT/f3 && grep scan_mfma $OUT/k_kernel_stats.csv && cat $OUT/f3_fetch_size_cfg3.txt
