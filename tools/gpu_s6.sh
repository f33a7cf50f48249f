#!/usr/bin/env bash
# round 3, session 6: full GPU suite with the quadrant backward form, C3 timing, C4 bench + kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > gpurun_out/s6_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s6_pytest.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/s6_pytest.log | tail -15
timeout 600 python tools/ab.py "" "bwd_quad_tiles=0" "" > gpurun_out/s6_ab_new.log 2>&1; tail -3 gpurun_out/s6_ab_new.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/s6_prof" -- python "$OLDPWD/bench.py" --config c4 --steps 10 --warmup 3 --spinup-steps 10 --no-cpu-baseline --no-extras > "$OLDPWD/gpurun_out/s6_prof_c4.json" 2> "$OLDPWD/gpurun_out/s6_prof.err")
f=$(find gpurun_out/s6_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/s6_c4_kernel_stats.csv && head -12 gpurun_out/s6_c4_kernel_stats.csv | cut -c1-60,150-230
rm -rf gpurun_out/s6_prof
timeout 600 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/s6_bench_c4.json 2> gpurun_out/s6_bench_c4.err; tail -c 1500 gpurun_out/s6_bench_c4.json
