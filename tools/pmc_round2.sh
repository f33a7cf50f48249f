#!/usr/bin/env bash
# PMC counter passes (separate runs, --kernel-trace only beside --pmc) for the bench workload, round 2.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTD="$PWD"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
run() { # name, counters...
  local name="$1"; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOTD/gpurun_out/pmc/$name" -- python "$ROOTD/bench.py" --steps 3 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-extras --no-stage-timers > "$ROOTD/gpurun_out/pmc/$name.json" 2> "$ROOTD/gpurun_out/pmc/$name.err")
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python tools/collect_traffic.py gpurun_out/pmc r02 > gpurun_out/pmc/traffic.json 2> gpurun_out/pmc/collect.err
python tools/collect_sq.py gpurun_out/pmc r02 > gpurun_out/pmc/sq.json 2>> gpurun_out/pmc/collect.err
cp profiles/r02_pmc_traffic.json profiles/r02_pmc_sq.json gpurun_out/ 2>/dev/null
ls gpurun_out/pmc
