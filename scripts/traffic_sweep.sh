#!/bin/bash
# traffic_sweep.sh: FETCH_SIZE of the cfg3 batch (own rocprofv3 --pmc pass each) under a few settings of the 4-wave scan,
# one box: what the work stealing and the pacing interval do to the HBM reads.  -> gpurun_out/traffic_sweep.txt
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/ts; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # label, then the environment is whatever the caller exported
  timeout -k 10 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/$1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-probe --no-host-probe > $O/$1.json 2> $O/$1.err &&
  { echo "== $1"; python3 scripts/pmc_summary.py $O/$1 "scan_mfma_w4_kernel<0, 0" | grep -E "dispatches|FETCH_SIZE"; } >> $GRAFT_REPO_ROOT/gpurun_out/traffic_sweep.txt; rm -rf $O/$1
}
rm -f gpurun_out/traffic_sweep.txt
run default &&
export VROD_DEBUG_W4_STEAL=0 && run steal_off && unset VROD_DEBUG_W4_STEAL &&
export VROD_DEBUG_PACE_KT=96 && run pace_96 &&
export VROD_DEBUG_PACE_KT=48 && run pace_48 &&
export VROD_DEBUG_PACE_KT=96 VROD_DEBUG_W4_STEAL=0 && run pace_96_steal_off
cat gpurun_out/traffic_sweep.txt
