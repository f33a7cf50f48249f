#!/bin/bash
# Repeat the GPU test tier N times on one box and keep only what failed (flakiness hunt: the reference's atomics make
# its gradients differ run to run, and the bars of tests/helpers.py are judged against those runs).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${TAG:-repeat}; mkdir -p $O
: > $O/repeat_summary.log
for i in $(seq 1 ${N:-8}); do
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/run_$i.log 2>&1
  echo "run $i: $(grep -a 'passed\|failed' $O/run_$i.log | tail -1)" >> $O/repeat_summary.log
  grep -a "^FAILED\|^E  " $O/run_$i.log | head -20 >> $O/repeat_summary.log
  if ! grep -aq "failed" $O/run_$i.log; then rm $O/run_$i.log; fi
done
cat $O/repeat_summary.log
