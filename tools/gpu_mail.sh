#!/usr/bin/env bash
# counter mailbox: its test, then bench lines with the option on / off (same box, alternating)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "mailbox or prefiltered or reproducible or radii_may or deferred or long_lists" > gpurun_out/mail_pytest.log 2>&1
tail -5 gpurun_out/mail_pytest.log
for cfg in c3 c2 c4; do
  for mb in 1 0 1 0; do
    timeout 600 python bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-stage-timers --option counter_mailbox=$mb 2> gpurun_out/mail_err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg mailbox=$mb', round(d['ms_per_step'],4), round(d['value'],1))"
  done
done 2>&1 | tee gpurun_out/mail_ab.log
