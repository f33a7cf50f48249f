#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "mailbox" > gpurun_out/mail_pytest.log 2>&1
tail -3 gpurun_out/mail_pytest.log
for cfg in c2 c3; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/tr" -- python "$OLDPWD/bench.py" --config $cfg --steps 10 --warmup 3 --spinup-steps 10 --no-cpu-baseline --no-extras --no-stage-timers > /dev/null 2> "$OLDPWD/gpurun_out/tr.err")
  f=$(find gpurun_out/tr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/mail_${cfg}_kernels.csv
  rm -rf gpurun_out/tr
done
