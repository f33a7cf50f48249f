#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; : > gpurun_out/quick2.log
for mode in "" "--deferred-counters"; do
  for i in 1 2; do
  echo "== bench no-timers $mode" >> gpurun_out/quick2.log
  timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-stage-timers $mode 2>&1 | grep -v "amdgpu.ids" | tail -1 | cut -c1-200 >> gpurun_out/quick2.log
  done
done
export TMPDIR=/tmp; ROOTD="$PWD"; rm -rf gpurun_out/prof_def
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$ROOTD/gpurun_out/prof_def" -- python "$ROOTD/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-stage-timers > /dev/null 2> "$ROOTD/gpurun_out/prof_def.err")
