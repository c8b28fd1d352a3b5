#!/bin/bash
# lib_traffic.sh "name1 name2 ...": per library variant (vrod_amd/libvrod_NAME.so), FETCH_SIZE of the cfg3 batch (own rocprofv3
# --pmc pass) -> gpurun_out/lib_traffic.txt
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/lt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/lib_traffic.txt
for n in $1; do
  export VROD_HIP_LIB=$GRAFT_REPO_ROOT/vrod_amd/libvrod_$n.so
  timeout -k 10 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-probe --no-host-probe > $O/$n.json 2> $O/$n.err &&
  { echo "== $n"; python3 scripts/pmc_summary.py $O/$n "scan_mfma_w4_kernel<0, 0" | grep -E "FETCH_SIZE"; } >> gpurun_out/lib_traffic.txt; rm -rf $O/$n
done
cat gpurun_out/lib_traffic.txt
