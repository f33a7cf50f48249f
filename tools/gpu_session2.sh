#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=frosting_amd/lib/libfrosting_rasterizer.so
cp $L /tmp/new.so
for rep in 1 2; do
cp frosting_amd/lib_alt/base.so $L
timeout 600 python -m pytest tests/test_gpu_mesh.py -m gpu -q -rA -k c4_refine 2>&1 | grep -E "^c4 |passed|failed" > gpurun_out/s2_c4_base_$rep.log
cp /tmp/new.so $L
timeout 600 python -m pytest tests/test_gpu_mesh.py -m gpu -q -rA -k c4_refine 2>&1 | grep -E "^c4 |passed|failed" > gpurun_out/s2_c4_new_$rep.log
done
tail -n 20 gpurun_out/s2_c4_*.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s2_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s2_pytest.log
grep -E "^FAILED|passed|failed" gpurun_out/s2_pytest.log | tail
