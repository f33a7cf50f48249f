#!/usr/bin/env bash
# GPU box: test tier + bench variants (deferred / blocking counters); logs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
: > gpurun_out/quick.log
for mode in "" "--deferred-counters"; do
  echo "== bench $mode" >> gpurun_out/quick.log
  timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $mode 2>&1 | grep -v "amdgpu.ids" | tail -2 >> gpurun_out/quick.log
done
timeout 120 python tools/profile_stages.py c3 2>&1 | grep -v "amdgpu.ids" >> gpurun_out/quick.log
grep -n "passed\|failed\|Error\|rc=" gpurun_out/pytest_gpu.log | head -5
