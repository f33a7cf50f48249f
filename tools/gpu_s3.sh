#!/usr/bin/env bash
# round 3, session 3: full GPU suite, A/B vs HEAD of round 2 (lib_alt/base.so), skew scene timeline, training step, bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > gpurun_out/s3_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s3_pytest.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/s3_pytest.log | tail -15
FROSTING_LIB=$PWD/frosting_amd/lib_alt/base.so timeout 600 python tools/ab.py "" > gpurun_out/s3_ab_base.log 2>&1; tail -1 gpurun_out/s3_ab_base.log
timeout 600 python tools/ab.py "" "" > gpurun_out/s3_ab_new.log 2>&1; tail -2 gpurun_out/s3_ab_new.log
timeout 600 python tools/ab.py --scene skew "" > gpurun_out/s3_skew_new.log 2>&1; tail -1 gpurun_out/s3_skew_new.log
timeout 600 python tools/train_step.py > gpurun_out/s3_train_step.log 2>&1; tail -4 gpurun_out/s3_train_step.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/s3_prof" -- python "$OLDPWD/tools/ab.py" --scene skew --steps 20 "" > "$OLDPWD/gpurun_out/s3_prof.log" 2>&1)
f=$(find gpurun_out/s3_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/s3_skew_kernel_stats.csv
f=$(find gpurun_out/s3_prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/s3_skew_kernel_trace.csv
rm -rf gpurun_out/s3_prof
