#!/usr/bin/env bash
# round 3, session 9: sort ranking A/B (prev.so = committed build), quadrant form batch 2 vs 3 on C4
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile_sort or depth_ties or golden or quadrant" > gpurun_out/s9_pytest.log 2>&1; tail -2 gpurun_out/s9_pytest.log
FROSTING_LIB=$PWD/frosting_amd/lib_alt/prev.so timeout 600 python tools/ab.py "" "" > gpurun_out/s9_ab_prev.log 2>&1; tail -2 gpurun_out/s9_ab_prev.log
timeout 600 python tools/ab.py "" "" > gpurun_out/s9_ab_new.log 2>&1; tail -2 gpurun_out/s9_ab_new.log
FROSTING_LIB=$PWD/frosting_amd/lib_alt/prev.so timeout 600 python tools/ab.py "" > gpurun_out/s9_ab_prev2.log 2>&1; tail -1 gpurun_out/s9_ab_prev2.log
timeout 600 python tools/ab.py "" > gpurun_out/s9_ab_new2.log 2>&1; tail -1 gpurun_out/s9_ab_new2.log
for b in 3 2; do
  timeout 600 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --option bwd_batch=$b > gpurun_out/s9_c4_b$b.json 2> gpurun_out/s9_c4_b$b.err
  python - "$b" <<'PY'
import json,sys
b=sys.argv[1]
d=json.loads(open(f'gpurun_out/s9_c4_b{b}.json').read().strip().splitlines()[-1])
print('c4 bwd_batch', b, d['ms_per_step'], d['stage_ms'])
PY
done
