#!/usr/bin/env bash
# TEST INFRASTRUCTURE: builds the *reference's own* rasterizer for gfx950 as a
# parity oracle / "reference on MI355X" timing baseline.
#
#   sources : /root/reference/gaussian_splatting/submodules/diff-gaussian-rasterization/
#             cuda_rasterizer/{rasterizer_impl,forward,backward}.cu  (compiled where they lie,
#             through a throw-away sed-fixed copy in $TMP: clang rejects the spaced
#             launch chevrons `<< <` / `>> >`; nothing is copied into this repo)
#   shims   : oracle/shims/  (cuda_runtime.h -> hip, cub -> hipcub, cooperative_groups)
#   outputs : oracle/_ref/libref_rasterizer_{exact,fast}.so   (git-ignored, ships via gpurun)
#             exact = -ffp-contract=off (deterministic IEEE op order; used for bit-exact
#                     integer/key parity), fast = hipcc default contraction (timing baseline
#                     and tolerance parity)
# The reference's own build system (setup.py / CMake) is not run.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${FROSTING_REFERENCE:-/root/reference}/gaussian_splatting/submodules/diff-gaussian-rasterization"
if [ ! -d "$REF/cuda_rasterizer" ]; then
  echo "build_ref.sh: reference sources not present at $REF -- skipping (prebuilt oracle/_ref is used if it exists)"
  exit 0
fi
TMP="$(mktemp -d /tmp/frosting_ref_build.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/cuda_rasterizer" "$HERE/_ref"
for f in "$REF"/cuda_rasterizer/*; do
  sed -e 's/<< </<<</g' -e 's/>> >/>>>/g' "$f" > "$TMP/cuda_rasterizer/$(basename "$f")"
done
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip
        -I"$HERE/shims" -I"$REF/third_party/glm" -I"$TMP/cuda_rasterizer"
        -Wno-unused-result -Wno-deprecated-declarations -w)
build() { # $1 = variant name, rest = extra flags
  local name="$1"; shift
  local objs=()
  for src in rasterizer_impl forward backward; do
    "$HIPCC" "${COMMON[@]}" "$@" -c "$TMP/cuda_rasterizer/$src.cu" -o "$TMP/${name}_$src.o" &
    objs+=("$TMP/${name}_$src.o")
  done
  "$HIPCC" "${COMMON[@]}" "$@" -c "$HERE/ref_wrapper.cpp" -o "$TMP/${name}_wrapper.o" &
  objs+=("$TMP/${name}_wrapper.o")
  wait
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$HERE/_ref/libref_rasterizer_${name}.so"
  echo "built $HERE/_ref/libref_rasterizer_${name}.so"
}
build exact -ffp-contract=off
build fast

# ---- simple-knn (distCUDA2): gaussian_splatting/submodules/simple-knn/simple_knn.cu, same recipe ----------
# hipCUB / rocThrust stand in for CUB / Thrust through the same shims; -ffp-contract=off so that the squared
# distances are the reference's left-to-right float32 sums (its own build leaves contraction to nvcc).
KNN="${FROSTING_REFERENCE:-/root/reference}/gaussian_splatting/submodules/simple-knn"
if [ -f "$KNN/simple_knn.cu" ]; then
  mkdir -p "$TMP/knn"
  sed -e 's/<< </<<</g' -e 's/>> >/>>>/g' "$KNN/simple_knn.cu" > "$TMP/knn/simple_knn.cu"
  "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -I"$HERE/shims" -I"$KNN" -w -ffp-contract=off \
      -c "$TMP/knn/simple_knn.cu" -o "$TMP/knn/simple_knn.o"
  "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -I"$HERE/shims" -I"$KNN" -w \
      -c "$HERE/ref_knn_wrapper.cpp" -o "$TMP/knn/wrapper.o"
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "$TMP/knn/simple_knn.o" "$TMP/knn/wrapper.o" -o "$HERE/_ref/libref_simple_knn.so"
  echo "built $HERE/_ref/libref_simple_knn.so"
fi
