#!/usr/bin/env bash
# round 3, session 10: priority side stream for the 16-wave per-Gaussian backward -- skew scene and C4, kernel timeline
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab.py --scene skew "" > gpurun_out/s10_skew.log 2>&1; tail -1 gpurun_out/s10_skew.log
timeout 600 python tools/ab.py "" > gpurun_out/s10_c3.log 2>&1; tail -1 gpurun_out/s10_c3.log
timeout 600 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/s10_c4.json 2> gpurun_out/s10_c4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s10_c4.json').read().strip().splitlines()[-1]); print('c4', d['ms_per_step'], d['stage_ms'])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/s10_prof" -- python "$OLDPWD/tools/ab.py" --scene skew --steps 20 "" > "$OLDPWD/gpurun_out/s10_prof.log" 2>&1)
f=$(find gpurun_out/s10_prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/s10_skew_kernel_trace.csv
f=$(find gpurun_out/s10_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/s10_skew_kernel_stats.csv
rm -rf gpurun_out/s10_prof
