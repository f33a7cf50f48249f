#!/usr/bin/env bash
# same-box A/B of one library option: tools/gpu_ab_opt.sh name   (values 1 / 0, alternating, C3 / C4 / skew)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
opt=$1
: > gpurun_out/ab_opt.log
for rep in 1 2 3; do
 for v in 1 0; do
  for cfg in c3 c4; do
    timeout 600 python bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-stage-timers --option $opt=$v 2>> gpurun_out/ab_opt.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$opt=$v $cfg', round(d['ms_per_step'],4))" >> gpurun_out/ab_opt.log
  done
 done
done
timeout 600 python tools/ab.py --scene skew --steps 30 "$opt=1" "$opt=0" "$opt=1" "$opt=0" 2>> gpurun_out/ab_opt.err | grep "ms/step" >> gpurun_out/ab_opt.log
sort gpurun_out/ab_opt.log
