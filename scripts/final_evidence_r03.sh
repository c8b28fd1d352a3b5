#!/bin/bash
# Round-3 end-of-round evidence on ONE box (one gpurun call): the default bench, a sustained run, the shard sizes of the
# 1 / 2 / 4 / 8-GPU split with the exchange in the step, the other configs, and per-launch timelines.  -> gpurun_out/final
set -o pipefail
O=gpurun_out/final; mkdir -p $O
B="--no-cpu-baseline --no-hbm-probe --no-host-probe"
python -c "import __graft_entry__ as g; g.smoke()" > $O/k_smoke.log 2>&1 &&
python bench.py --steps 20 --warmup 5 > $O/k_bench_default.json 2> $O/k_bench_default.err &&
python bench.py --steps 400 --warmup 5 $B > $O/k_sustained_400_batches.json 2> $O/s.err &&
VROD_BENCH_FORCE_COLLECTIVE=1 VROD_BENCH_VERIFY=0 python bench.py --rows 1250000 --steps 200 --warmup 10 $B > $O/k_shard_1p25M_with_exchange.json 2> $O/sh.err &&
python bench.py --rows 1250000 --steps 200 --warmup 10 $B > $O/k_shard_1p25M.json 2> $O/sh1.err &&
python bench.py --rows 2500000 --steps 100 --warmup 10 $B > $O/k_shard_2p5M.json 2> $O/sh2.err &&
python bench.py --rows 5000000 --steps 60 --warmup 5 $B > $O/k_shard_5M.json 2> $O/sh5.err &&
python bench.py --workload cfg2 --steps 60 --warmup 5 --no-cpu-baseline > $O/k_bench_cfg2.json 2> $O/c2.err &&
python bench.py --workload cfg5 --steps 8 --warmup 2 --no-cpu-baseline > $O/k_bench_cfg5_split.json 2> $O/c5.err &&
python bench.py --rows 40000000 --steps 8 --warmup 2 $B > $O/k_cfg4_on_one_gpu_40M.json 2> $O/c4.err &&
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT &&
rocprofv3 --kernel-trace --output-format csv -d $O/t10 -- python3 scripts/ab_probe.py 10000000 1024 4 > $O/t10.log 2>&1 &&
python scripts/step_timeline.py $O/t10 > $O/k_step_timeline_10M.txt &&
rocprofv3 --kernel-trace --output-format csv -d $O/t1 -- python3 scripts/ab_probe.py 1250000 1024 6 > $O/t1.log 2>&1 &&
python scripts/step_timeline.py $O/t1 > $O/k_step_timeline_1p25M_shard.txt &&
rocprofv3 --kernel-trace --output-format csv -d $O/tp -- python3 bench.py --steps 12 --warmup 3 --rows 1250000 $B > $O/tp.json 2> $O/tp.err &&
python scripts/pipeline_timeline.py $O/tp 2 > $O/s_pipeline_timeline_shard.txt &&
python scripts/probes/k_probe.py 8000000 1024 768 bf16 "10 100 1000" 2>&1 | grep -v amdgpu.ids > $O/k_probe_k_8Mx768_q1024.txt &&
rm -rf $O/t10 $O/t1 $O/tp &&
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final/k_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('launches_per_step'))
PY
