#!/bin/bash
# run_lib_ab.sh "lib names" [rows] ["thr list"]: scripts/ubench/lib_ab over vrod_amd/libvrod_NAME.so at several hit densities
cd "$(dirname "$0")/../.."
libs=""; for n in $1; do libs="$libs $PWD/vrod_amd/libvrod_$n.so"; done
for thr in ${3:-1e30 0.16 0.145 0.13}; do ./scripts/ubench/lib_ab ${2:-4194304} 8 5 $thr $libs; done
