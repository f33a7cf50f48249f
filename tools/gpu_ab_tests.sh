#!/usr/bin/env bash
# tests on the current build, then A/B of the previous (lib_alt/base.so) and current library
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > gpurun_out/abt_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/abt_pytest.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/abt_pytest.log | tail -15
FROSTING_LIB=$PWD/frosting_amd/lib_alt/base.so timeout 600 python tools/ab.py "" "tight_binning=1" > gpurun_out/abt_ab_base.log 2>&1
tail -2 gpurun_out/abt_ab_base.log
timeout 600 python tools/ab.py "" "tight_binning=1" ${AB_EXTRA:-} "" > gpurun_out/abt_ab_new.log 2>&1
tail -4 gpurun_out/abt_ab_new.log
if [ -n "${PROF:-}" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/abt_prof" -- python "$OLDPWD/tools/ab.py" --steps 20 "" > "$OLDPWD/gpurun_out/abt_prof.log" 2>&1)
  f=$(find gpurun_out/abt_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/abt_kernel_stats.csv && head -16 gpurun_out/abt_kernel_stats.csv | cut -c1-150
fi
