#!/bin/bash
# build everything, then hand the command to gpurun (the built .so files travel with the snapshot)
set -e
cd "$(dirname "$0")/.."
make -C vrod_amd/csrc -j6 >/dev/null
make -C vrod_amd/host >/dev/null
make -C oracle >/dev/null
exec /usr/local/graft/bin/gpurun "$@"
