#!/usr/bin/env bash
# A/B builds of the HIP library: one per "name=-DSWITCH [-DSWITCH...]" argument, into frosting_amd/lib_ab/<name>/ (git-ignored;
# travels with a gpurun snapshot; select with FROSTING_LIB=$PWD/frosting_amd/lib_ab/<name>/libfrosting_rasterizer.so).
# `tools/build_variants.sh clean` removes them -- none may be left when a round ends.
#   tools/build_variants.sh rcp=-DFRG_AB_RCP power=-DFRG_AB_POWER exp=-DFRG_AB_EXP nofma=-DFRG_AB_NOFMA
set -e
cd "$(dirname "$0")/.."
if [ "${1:-}" = clean ]; then rm -rf frosting_amd/lib_ab build/ab_*; exit 0; fi
for spec in "$@"; do
  name="${spec%%=*}"; defs="${spec#*=}"
  make -C frosting_amd/csrc -j8 BUILD=../../build/ab_$name OUTDIR=../lib_ab/$name DEFS="$defs" > /tmp/build_ab_$name.log 2>&1 \
    || { grep -E "error" -A4 /tmp/build_ab_$name.log | head -30; echo "variant $name FAILED"; exit 1; }
  echo "built frosting_amd/lib_ab/$name/libfrosting_rasterizer.so ($defs)"
done
