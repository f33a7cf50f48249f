#!/usr/bin/env bash
# A/B of library builds given as LIBS="a.so b.so ..." (paths relative to the repo), tools/ab.py default settings
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for rep in 1 2; do
for l in ${LIBS}; do
  echo "== $l (pass $rep)"
  FROSTING_LIB=$PWD/$l timeout 600 python tools/ab.py "" ${AB_EXTRA:-} 2>&1 | grep "^\[" | cut -c1-220
done
done | tee gpurun_out/abl_ab.log
