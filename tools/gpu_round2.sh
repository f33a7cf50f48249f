#!/usr/bin/env bash
# One gpurun call of round 2: GPU test tier (all tests, no -x, durations), smoke, bench, rocprof kernel stats.
# STEPS to run: env WHAT="tests smoke bench prof" (default all)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
WHAT="${WHAT:-tests smoke bench prof}"
export TMPDIR=/tmp
ROOTD="$PWD"
for w in $WHAT; do
case $w in
tests)
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q -rA --durations=15 ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  grep -E "passed|failed|FAILED|ERROR|rc=" gpurun_out/pytest_gpu.log | tail -30 ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
  tail -2 gpurun_out/smoke.log ;;
bench)
  timeout 900 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
  tail -c 6000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err ;;
prof)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTD/gpurun_out/prof" -- python "$ROOTD/bench.py" --steps 10 --warmup 3 --spinup-steps 20 --no-cpu-baseline --no-extras > "$ROOTD/gpurun_out/prof_bench.json" 2> "$ROOTD/gpurun_out/prof.err")
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv && head -25 gpurun_out/kernel_stats.csv ;;
esac
done
