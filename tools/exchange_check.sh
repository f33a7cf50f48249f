#!/usr/bin/env bash
# GPU box: test tier + single-rank RCCL runs of both exchange plans (tools only; results under gpurun_out/).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
: > gpurun_out/exchange.log
for ex in allreduce factored; do
  for mode in "--sync-exchange" ""; do
    echo "== $ex $mode" >> gpurun_out/exchange.log
    timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-exchange --exchange $ex $mode 2>&1 | grep -v "^Librccl\|^RCCL\|^HIP\|^ROCm\|^Hostname" | tail -3 >> gpurun_out/exchange.log
  done
done
echo "== plain" >> gpurun_out/exchange.log
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> gpurun_out/exchange.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
