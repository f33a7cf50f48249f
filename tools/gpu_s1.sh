#!/usr/bin/env bash
# round 3, session 1: new sort / heavy-wave paths against the reference, then A/B of HEAD (lib_alt/base.so) and the new build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "long_lists or tile_sort or depth_ties or skewed" -rA > gpurun_out/s1_new_tests.log 2>&1; echo "new tests rc=$?" | tee -a gpurun_out/s1_new_tests.log
grep -E "gradient rel-L2|passed|failed|FAILED|ERROR|Error|assert" gpurun_out/s1_new_tests.log | cut -c1-400 | tail -30
timeout 1200 python -m pytest tests -m gpu -q -rA --durations=10 -k "not (long_lists or tile_sort or depth_ties or skewed)" > gpurun_out/s1_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s1_pytest.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/s1_pytest.log | tail -15
FROSTING_LIB=$PWD/frosting_amd/lib_alt/base.so timeout 600 python tools/ab.py "" > gpurun_out/s1_ab_base.log 2>&1; tail -1 gpurun_out/s1_ab_base.log
timeout 600 python tools/ab.py "" "" > gpurun_out/s1_ab_new.log 2>&1; tail -2 gpurun_out/s1_ab_new.log
timeout 600 python tools/ab.py --deferred "" > gpurun_out/s1_ab_deferred.log 2>&1; tail -1 gpurun_out/s1_ab_deferred.log
FROSTING_LIB=$PWD/frosting_amd/lib_alt/base.so timeout 600 python tools/ab.py --scene skew "" > gpurun_out/s1_skew_base.log 2>&1; tail -1 gpurun_out/s1_skew_base.log
timeout 600 python tools/ab.py --scene skew "" > gpurun_out/s1_skew_new.log 2>&1; tail -1 gpurun_out/s1_skew_new.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/s1_prof" -- python "$OLDPWD/tools/ab.py" --scene skew --steps 20 "" > "$OLDPWD/gpurun_out/s1_prof.log" 2>&1)
f=$(find gpurun_out/s1_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/s1_skew_kernel_stats.csv && head -24 gpurun_out/s1_skew_kernel_stats.csv | cut -c1-60,200-330
