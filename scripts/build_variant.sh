#!/bin/bash
# build_variant.sh NAME [extra hipcc flags for kernels_mfma_w4.hip]: an A/B copy of the library,
# vrod_amd/libvrod_NAME.so (git-ignored; select it with VROD_HIP_LIB), from the current sources.
# It is the product Makefile with another output and object directory, so the register audit of the
# asm-owned accumulator file (scripts/audit_w4.py) gates it exactly as it gates the product build.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
rm -f vrod_amd/libvrod_$name.so
make -C vrod_amd/csrc -j8 OUT=../libvrod_$name.so OBJDIR=/tmp/variant_$name/obj W4FLAGS="$*" >/tmp/variant_$name.log 2>&1 \
  || { tail -8 /tmp/variant_$name.log; echo "variant $name NOT built"; exit 1; }
echo "built vrod_amd/libvrod_$name.so (audit ok)"
