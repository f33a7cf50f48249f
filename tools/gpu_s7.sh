#!/usr/bin/env bash
# round 3, session 7: quadrant-form tests, C4 with the quadrant form forced / never / automatic
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rA -k "quadrant or deferred or golden or c_oracle or reproducible" > gpurun_out/s7_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s7_pytest.log
grep -E "passed|failed|^FAILED|^ERROR|quadrant form vs" gpurun_out/s7_pytest.log | cut -c1-260 | tail -30
for q in 0 100000 -1; do
  FROSTING_BWD_QUAD_TILES=$q timeout 600 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/s7_c4_q$q.json 2> gpurun_out/s7_c4_q$q.err
  python - "$q" <<'PY'
import json,sys
q=sys.argv[1]
d=json.loads(open(f'gpurun_out/s7_c4_q{q}.json').read().strip().splitlines()[-1])
print('c4 quad_tiles', q, d['ms_per_step'], d['stage_ms'])
PY
done
FROSTING_BWD_QUAD_TILES=100000 timeout 600 python tools/ab.py --steps 20 "" > gpurun_out/s7_c3_quad.log 2>&1; tail -1 gpurun_out/s7_c3_quad.log
