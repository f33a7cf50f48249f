#!/usr/bin/env bash
# same-box A/B of frosting_amd/lib_alt/base.so (the previous commit's library) against the current build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/ab_lib.log
for rep in 1 2; do
 for which in base cur; do
  if [ $which = base ]; then export FROSTING_LIB=$PWD/frosting_amd/lib_alt/base.so; else unset FROSTING_LIB; fi
  for cfg in c3 c2 c4; do
    timeout 600 python bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-stage-timers 2>> gpurun_out/ab_lib.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which $cfg', round(d['ms_per_step'],4))" >> gpurun_out/ab_lib.log
  done
  timeout 600 python tools/ab.py --scene skew --steps 30 "" 2>> gpurun_out/ab_lib.err | grep -i "wall\|ms/step\|step" | tail -1 | sed "s/^/$which skew /" >> gpurun_out/ab_lib.log
 done
done
cat gpurun_out/ab_lib.log
