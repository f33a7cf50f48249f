#!/usr/bin/env bash
# Same-box A/B of library builds and options: for every SETTING ("name|FROSTING_LIB path or -|bench.py args") two alternating
# passes of bench.py --no-extras on CONFIGS (default "c3"); prints ms per step and the per-stage hipEvent times.
# usage (through gpurun): SETTINGS="base|frosting_amd/lib_alt/base.so| cur|-| b2|-|--option bwd_batch=2" CONFIGS="c3 c4" bash tools/gpu_ab4.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04
LOG=gpurun_out/r04/ab4.log
: > $LOG
for rep in 1 2; do
 for setting in ${SETTINGS}; do
  name="${setting%%|*}"; rest="${setting#*|}"; lib="${rest%%|*}"; extra="${rest#*|}"; extra="${extra//,/ }"
  if [ "$lib" != "-" ]; then export FROSTING_LIB="$PWD/$lib"; else unset FROSTING_LIB; fi
  for cfg in ${CONFIGS:-c3}; do
    timeout 600 python bench.py --config $cfg --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --no-extras $extra 2>> gpurun_out/r04/ab4.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %-3s %.4f ms/step | ' % ('$name', '$cfg', d['ms_per_step']) + ' '.join('%s %.3f' % (k, v) for k, v in d.get('stage_ms', {}).items()))" >> $LOG
  done
 done
done
cat $LOG
