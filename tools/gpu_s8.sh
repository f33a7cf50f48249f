#!/usr/bin/env bash
# round 3, session 8: crossover of the two backward-blend forms on scenes covering part of the image
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for sh in 0.45 0.6 0.75; do
  timeout 600 python tools/ab.py --shrink $sh --steps 20 "bwd_quad_tiles=0" "bwd_quad_tiles=100000" > gpurun_out/s8_shrink_$sh.log 2>&1
  echo "shrink $sh"; grep -E "^R |^\[" gpurun_out/s8_shrink_$sh.log | cut -c1-250
done
