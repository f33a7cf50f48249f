#!/usr/bin/env bash
# PMC counter passes (separate runs, kernel-trace only) for the bench workload.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTD="$PWD"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
run() { # name, counters...
  local name="$1"; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOTD/gpurun_out/pmc/$name" -- python "$ROOTD/bench.py" --steps 3 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-tight-pass --no-stage-timers > "$ROOTD/gpurun_out/pmc/$name.json" 2> "$ROOTD/gpurun_out/pmc/$name.err")
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
find gpurun_out/pmc -name "*counter_collection.csv" | head; ls gpurun_out/pmc
