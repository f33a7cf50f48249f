#!/usr/bin/env bash
# round 3, session 2: long-list / heavy-wave tests vs the reference, A/B of HEAD (lib_alt/base.so) and the new build, bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "long_lists or tile_sort or depth_ties or skewed" -rA > gpurun_out/s2_new_tests.log 2>&1; echo "new tests rc=$?" | tee -a gpurun_out/s2_new_tests.log
grep -E "gradient rel-L2|passed|failed|^FAILED|^ERROR|Error|assert " gpurun_out/s2_new_tests.log | cut -c1-420 | tail -30
FROSTING_LIB=$PWD/frosting_amd/lib_alt/base.so timeout 600 python tools/ab.py "" > gpurun_out/s2_ab_base.log 2>&1; tail -1 gpurun_out/s2_ab_base.log
timeout 600 python tools/ab.py "" "" > gpurun_out/s2_ab_new.log 2>&1; tail -2 gpurun_out/s2_ab_new.log
timeout 600 python tools/ab.py --scene skew "" > gpurun_out/s2_skew_new.log 2>&1; tail -1 gpurun_out/s2_skew_new.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/s2_bench.json').read().strip().splitlines()[-1])
    print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step')}, d.get('roofline',{}).get('frac'), d.get('stage_ms'))
    for k in ('c2','c4','skew_scene','api_path','tight_binning'):
        v=d.get(k); print(k, {kk:vv for kk,vv in v.items() if kk in ('ms_per_step','frac','mesh_raster_ms','error','stage_ms','vs_c_abi')} if v else None)
    print(d['op_hbm'])
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/s2_bench.err').read()[-2000:])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/s2_prof" -- python "$OLDPWD/tools/ab.py" --scene skew --steps 20 "" > "$OLDPWD/gpurun_out/s2_prof.log" 2>&1)
f=$(find gpurun_out/s2_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/s2_skew_kernel_stats.csv
