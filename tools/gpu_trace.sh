#!/usr/bin/env bash
# kernel timelines (rocprofv3 --kernel-trace) of the small configs: where do C2 (forward, 100 k Gaussians) and C4 spend a step?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in c2 c4; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/trace_$cfg" -- python "$OLDPWD/bench.py" --config $cfg --steps 10 --warmup 3 --spinup-steps 10 --no-cpu-baseline --no-extras --no-stage-timers > "$OLDPWD/gpurun_out/trace_$cfg.json" 2> "$OLDPWD/gpurun_out/trace_$cfg.err")
  f=$(find gpurun_out/trace_$cfg -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/trace_${cfg}_kernels.csv
  f=$(find gpurun_out/trace_$cfg -name "*memory_copy_trace.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/trace_${cfg}_copies.csv
  rm -rf gpurun_out/trace_$cfg
done
ls -la gpurun_out/trace_*
