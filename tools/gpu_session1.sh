#!/usr/bin/env bash
# Round-2 diagnostic session: GPU tests on the current build, then A/B timings (tools/ab.py) of the previous build
# (frosting_amd/lib_alt/base.so) and the current one, the stage ablations, the overlap probes and the effect of the
# memory order of the caller's Gaussians on the binning stages.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/s1_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s1_pytest.log
tail -3 gpurun_out/s1_pytest.log
FROSTING_LIB=$PWD/frosting_amd/lib_alt/base.so timeout 600 python tools/ab.py "" "tight_binning=1" "ablate=1" "ablate=2" "ablate=3" "" > gpurun_out/s1_ab_base.log 2>&1
cat gpurun_out/s1_ab_base.log | tail -8
timeout 600 python tools/ab.py "" "probe=1" "probe=2" "probe=3" "tight_binning=1" "" > gpurun_out/s1_ab_new.log 2>&1
tail -8 gpurun_out/s1_ab_new.log
timeout 600 python tools/ab.py --order yrow "" "tight_binning=1" > gpurun_out/s1_ab_yrow.log 2>&1
tail -3 gpurun_out/s1_ab_yrow.log
timeout 600 python tools/ab.py --order morton "" > gpurun_out/s1_ab_morton.log 2>&1
tail -2 gpurun_out/s1_ab_morton.log
