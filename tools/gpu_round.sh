#!/usr/bin/env bash
# One gpurun call: golden fixtures from the reference, GPU test tier, smoke, bench, rocprof.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
if [ "${MAKE_GOLDEN:-1}" = "1" ]; then
  python tools/make_golden.py > gpurun_out/golden.log 2>&1 && cp gpurun_out/golden/*.npz tests/golden/
fi
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
export TMPDIR=/tmp
ROOTD="$PWD"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof" -- python "$ROOTD/bench.py" --steps 10 --warmup 3 --spinup-steps 20 --no-cpu-baseline --no-tight-pass > "$ROOTD/gpurun_out/prof_bench.json" 2> "$ROOTD/gpurun_out/prof.err")
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -2; cat gpurun_out/bench.json
find gpurun_out/prof -name "*stats*" | head
