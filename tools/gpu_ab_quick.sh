#!/usr/bin/env bash
# quick A/B + kernel stats of the current build (no tests)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab.py "" ${AB_EXTRA:-} "" > gpurun_out/abq_ab_new.log 2>&1
grep "^\[" gpurun_out/abq_ab_new.log | cut -c1-220
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/abq_prof" -- python "$OLDPWD/tools/ab.py" --steps 20 "" > "$OLDPWD/gpurun_out/abq_prof.log" 2>&1)
f=$(find gpurun_out/abq_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/abq_kernel_stats.csv
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/abq_kernel_stats.csv')):
    if 'frg::' in r['Name']:
        print(f"{r['Name'][:58]:58s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
