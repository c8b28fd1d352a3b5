#!/bin/bash
# Round-3 rocprofv3 evidence, one gpurun call.  Counter passes are separate from trace passes (MI355X_MICROARCH.md,
# rocprofv3 PMC slots), each counter list fits its block's slots, the program follows `--` directly.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
RP="rocprofv3 --output-format csv"
T="timeout -k 10"
step() { echo "== $1" | tee -a $OUT/progress.log; }
# 1. the default bench under --kernel-trace --stats (the JSON line's avg_launch_ms must agree with the stats)
step "k: kernel stats of the default bench" &&
$T 900 $RP --kernel-trace --stats -d $OUT/k -- python3 bench.py --steps 20 --warmup 5 > $OUT/k_bench_under_rocprof.json 2> $OUT/k.err &&
cp $(ls $OUT/k/*/*kernel_stats.csv | head -1) $OUT/k_kernel_stats.csv &&
# 2. FETCH_SIZE of cfg3 (own pass)
step "f3: FETCH_SIZE cfg3" &&
$T 600 $RP --pmc FETCH_SIZE -d $OUT/f3 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-probe --no-host-probe > $OUT/f3_bench.json 2> $OUT/f3.err &&
python3 scripts/pmc_summary.py $OUT/f3 scan_ > $OUT/f3_fetch_size_cfg3.txt &&
# 3. cfg2: stats + FETCH_SIZE
step "k2: cfg2 stats" &&
$T 600 $RP --kernel-trace --stats -d $OUT/k2 -- python3 bench.py --workload cfg2 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/k2_bench_cfg2.json 2> $OUT/k2.err &&
cp $(ls $OUT/k2/*/*kernel_stats.csv | head -1) $OUT/k2_kernel_stats_cfg2.csv &&
step "f2: cfg2 FETCH_SIZE" &&
$T 600 $RP --pmc FETCH_SIZE -d $OUT/f2 -- python3 bench.py --workload cfg2 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/f2_bench.json 2> $OUT/f2.err &&
python3 scripts/pmc_summary.py $OUT/f2 scan_ > $OUT/f2_fetch_size_cfg2.txt &&
# 4. SQ counters of the batched scan (2M rows)
step "sq: SQ pass" &&
$T 300 $RP --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/sq -- python3 scripts/ab_probe.py 2000000 1024 3 > $OUT/sq.log 2> $OUT/sq.err &&
python3 scripts/pmc_summary.py $OUT/sq scan_ > $OUT/sq_pmc_sq.txt &&
# 6. cfg5: split pass and fp32 pass
step "k5: cfg5 split stats" &&
$T 900 $RP --kernel-trace --stats -d $OUT/k5 -- python3 bench.py --workload cfg5 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/k5_bench_cfg5_split.json 2> $OUT/k5.err &&
cp $(ls $OUT/k5/*/*kernel_stats.csv | head -1) $OUT/k5_kernel_stats_cfg5_split.csv &&
step "f5: cfg5 split FETCH_SIZE" &&
$T 900 $RP --pmc FETCH_SIZE -d $OUT/f5 -- python3 bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/f5_bench.json 2> $OUT/f5.err &&
python3 scripts/pmc_summary.py $OUT/f5 scan_ > $OUT/f5_fetch_size_cfg5_split.txt &&
step "done"
# raw traces are large: keep the summaries only
rm -rf $OUT/k $OUT/f3 $OUT/k2 $OUT/f2 $OUT/sq $OUT/k5 $OUT/f5
ls -la $OUT
