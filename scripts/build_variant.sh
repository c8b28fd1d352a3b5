#!/bin/bash
# build_variant.sh NAME [extra hipcc flags for kernels_mfma.hip]: an A/B copy of the library,
# vrod_amd/libvrod_NAME.so (git-ignored; select it with VROD_HIP_LIB), from the current sources;
# the other objects are taken from build/obj.  The register audit of the asm-owned accumulator file
# (scripts/audit_w4.py) runs on the same flags: no library is left behind when it fails -- a variant
# that lets hipcc into the AGPRs faults on the GPU.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=/tmp/variant_$name
rm -rf $d; mkdir -p $d/obj; rm -f vrod_amd/libvrod_$name.so
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude -Ivrod_amd/csrc"
hipcc $F "$@" -S --cuda-device-only -o $d/k.s vrod_amd/csrc/kernels_mfma.hip 2>/dev/null &
hipcc $F "$@" -c vrod_amd/csrc/kernels_mfma.hip -o $d/obj/kernels_mfma.o
wait
python scripts/audit_w4.py $d/k.s > $d/audit.log || { grep -v -- "-> ok" $d/audit.log | tail -5; echo "AUDIT FAILED: $name not built"; exit 1; }
for o in build/obj/*.o; do [ "$(basename $o)" = kernels_mfma.o ] || cp $o $d/obj/; done
hipcc --offload-arch=gfx950 -shared -fPIC -o vrod_amd/libvrod_$name.so $d/obj/*.o
echo "built vrod_amd/libvrod_$name.so (audit ok)"
