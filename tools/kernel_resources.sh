#!/usr/bin/env bash
# Register / LDS / spill figures of every kernel of one HIP translation unit, compiled device-only
# for gfx950 with the flags the Makefile uses for it; --asm also leaves the disassembly in $OUT.
# usage: tools/kernel_resources.sh [--asm] <file.hip> [extra hipcc flags]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LLVM=/opt/rocm/lib/llvm/bin
ASM=0; if [ "${1:-}" = "--asm" ]; then ASM=1; shift; fi
src="$1"; shift
OUT="${OUT:-/tmp/kres}"; mkdir -p "$OUT"
base="$(basename "$src" .hip)"
extra=()
case "$base" in
  preprocess|preprocess_bwd|view_exchange|slot_exchange|adam) extra=(-ffp-contract=off) ;;
  blend_exact) extra=(-ffp-contract=off -fno-slp-vectorize) ;;
  blend_fast) extra=(-ffp-contract=off -fno-slp-vectorize) ;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I"$ROOT/frosting_amd/csrc" -I"$ROOT/include" \
    "${extra[@]}" "$@" --cuda-device-only --no-gpu-bundle-output -c "$src" -o "$OUT/$base.co"
$LLVM/llvm-readelf --notes "$OUT/$base.co" | python3 "$ROOT/tools/_parse_notes.py" "$base"
if [ $ASM = 1 ]; then $LLVM/llvm-objdump -d --no-show-raw-insn "$OUT/$base.co" > "$OUT/$base.s"; echo "disassembly: $OUT/$base.s"; fi
