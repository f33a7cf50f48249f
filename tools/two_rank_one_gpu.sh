#!/usr/bin/env bash
# Functional test of bench.py's multi-rank path on a one-GPU box: two ranks share GPU 0 and exchange
# through gloo (RCCL refuses two ranks on one device).  Numbers are meaningless; it must run clean.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; : > gpurun_out/two_rank.log
for ex in factored allreduce; do
  for mode in "" "--sync-exchange"; do
    echo "== $ex $mode" >> gpurun_out/two_rank.log
    FRG_BENCH_ONE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 2 --steps 6 --warmup 3 --spinup-steps 2 --backend gloo --exchange $ex $mode --points 400000 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^\*\*\*\|OMP_NUM" | tail -4 | cut -c1-700 >> gpurun_out/two_rank.log
  done
done
