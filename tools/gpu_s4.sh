#!/usr/bin/env bash
# round 3, session 4: full GPU suite, C3 / skew timings, bench line, C4 kernel profile
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 > gpurun_out/s4_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s4_pytest.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/s4_pytest.log | tail -15
FROSTING_LIB=$PWD/frosting_amd/lib_alt/base.so timeout 600 python tools/ab.py "" > gpurun_out/s4_ab_base.log 2>&1; tail -1 gpurun_out/s4_ab_base.log
timeout 600 python tools/ab.py "" "" > gpurun_out/s4_ab_new.log 2>&1; tail -2 gpurun_out/s4_ab_new.log
timeout 600 python tools/ab.py --scene skew "" "" > gpurun_out/s4_skew_new.log 2>&1; tail -2 gpurun_out/s4_skew_new.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s4_bench.json 2> gpurun_out/s4_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/s4_bench.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step')}, d.get('roofline',{}).get('frac'), d.get('stage_ms'))
    for k in ('c2','c4','skew_scene','api_path','tight_binning'):
        v=d.get(k); print(k, {kk:vv for kk,vv in v.items() if kk in ('ms_per_step','frac','mesh_raster_ms','error','stage_ms','vs_c_abi')} if v else None)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/s4_bench.err').read()[-2000:])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/s4_prof" -- python "$OLDPWD/bench.py" --config c4 --steps 10 --warmup 3 --spinup-steps 10 --no-cpu-baseline --no-extras > "$OLDPWD/gpurun_out/s4_prof_c4.json" 2> "$OLDPWD/gpurun_out/s4_prof.err")
f=$(find gpurun_out/s4_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/s4_c4_kernel_stats.csv && head -30 gpurun_out/s4_c4_kernel_stats.csv | cut -c1-70,150-260
rm -rf gpurun_out/s4_prof
