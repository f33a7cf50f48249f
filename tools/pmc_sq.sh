#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTD="$PWD"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc2
run() { local name="$1"; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOTD/gpurun_out/pmc2/$name" -- python "$ROOTD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$ROOTD/gpurun_out/pmc2/$name.json" 2> "$ROOTD/gpurun_out/pmc2/$name.err"); }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES
