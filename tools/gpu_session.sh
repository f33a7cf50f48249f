#!/usr/bin/env bash
# One gpurun call: `gpurun --timeout S -- 'WHAT="tests bench" TAG=r06 bash tools/gpu_session.sh'`.  The ONE GPU-session script
# (rounds 1-5 each had their own gpu_round*.sh / gpu_ab*.sh / pmc_round*.sh: folded into the modes below).
# WHAT selects the parts (default: tests sparse smoke bench prof); output under gpurun_out/$TAG.
#   tests     pytest -m gpu                                   sparse   tools/sparse_grad_check.py (gradients vs reference AND float64)
#   smoke     __graft_entry__.smoke()                         bench    the driver's bench line
#   prof      rocprofv3 --kernel-trace --stats of bench.py    ab       tools/ab.py with $AB_ARGS (one line per setting)
#   train     tools/train_step.py                             exchange single-rank exchange schedules
#   pmc       PMC passes LAST (FETCH/WRITE/SQ), collected into profiles/${TAG}_pmc_*.json with the build fingerprint
#   trace     kernel timeline of one step (TRACE_CFGS="c2 c3 c4")     gradab   default-arithmetic A/B builds vs float64
#   ab4/ab4b  alternating bench.py passes of library builds / options (SETTINGS, CONFIGS)
#   tworank   bench.py's multi-rank path on one GPU: two ranks share GPU 0 over gloo (functional only)
#   cmd       run $CMD (a one-off measurement) with its output in $O/cmd.log
#   combine   tools/combine_bench.py ($COMBINE_ARGS, "|"-separated; COMBINE_PROF=1: + rocprofv3 stats)      combinepmc   its PMC passes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${TAG:-r06}"
mkdir -p gpurun_out/$TAG
WHAT="${WHAT:-tests sparse smoke bench prof}"
export TMPDIR=/tmp
ROOTD="$PWD"; O="$ROOTD/gpurun_out/$TAG"
for w in $WHAT; do
case $w in
tests)
  eval "timeout 1800 python -m pytest ${PYTEST_PATHS:-tests} -m gpu -q -rA --durations=15 ${PYTEST_ARGS:-}" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  grep -E "passed|failed|^FAILED|^ERROR|rc=" $O/pytest_gpu.log | tail -15 ;;
sparse)
  FROSTING_EXPERIMENTS=1 timeout 1500 python tools/sparse_grad_check.py ${SPARSE_ARGS:-} > $O/sparse_grad_check.log 2>&1; echo "sparse rc=$?" >> $O/sparse_grad_check.log
  grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/sparse_grad_check.log | tail -120 ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log ;;
bench)
  timeout 1200 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:-} > $O/bench_c3.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench_c3.json; tail -3 $O/bench.err ;;
prof)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -- python "$ROOTD/bench.py" --steps 16 --warmup 3 --spinup-steps 24 --no-cpu-baseline --no-extras > "$O/prof_bench_c3.json" 2> "$O/prof.err")
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_c3_kernel_stats.csv && head -18 $O/bench_c3_kernel_stats.csv | cut -c1-70,150-230
  rm -rf $O/prof ;;
ab)
  FROSTING_EXPERIMENTS=1 timeout 900 python tools/ab.py ${AB_ARGS:-} > $O/ab.log 2>&1; echo "ab rc=$?" >> $O/ab.log; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/ab.log | tail -40 ;;
trace)
  # kernel timeline of one step of each of ${TRACE_CFGS:-c2} (gaps between launches): rocprofv3 kernel trace, no counters
  for tc in ${TRACE_CFGS:-${TRACE_CFG:-c2}}; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- python "$ROOTD/bench.py" --config $tc --steps 30 --warmup 5 --spinup-steps 50 --views 1 --no-cpu-baseline --no-extras --no-stage-timers ${TRACE_ARGS:-} > "$O/trace_bench.json" 2> "$O/trace.err")
    f=$(find $O/trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_gaps.py "$f" ${TRACE_PICK:-} > $O/trace_$tc.log 2>&1; cat $O/trace_$tc.log | tail -40
    rm -rf $O/trace
  done ;;
train)
  timeout 600 python tools/train_step.py > $O/train_step.log 2>&1; tail -6 $O/train_step.log ;;
exchange)
  : > $O/exchange_1rank.log
  IFS='|' read -ra EMODES <<< "${EXCHANGE_MODES:---exchange slotsum|--exchange slotsum --chunks 1|--exchange slotsum --chunks 2|--exchange slotsum --chunks 4|--exchange factored|--exchange factored --sync-exchange|--exchange factored --reduce direct|--exchange allreduce|--exchange sparse}"
  for mode in "${EMODES[@]}"; do
    echo "== bench.py --force-exchange $mode" >> $O/exchange_1rank.log
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --force-exchange $mode 2>&1 | grep -v "^Librccl\|^RCCL\|^HIP\|^ROCm\|^Hostname\|amdgpu.ids" | tail -2 >> $O/exchange_1rank.log
  done
  grep -c metric $O/exchange_1rank.log ;;
pmc)
  mkdir -p gpurun_out/pmc
  run() { local name="$1"; shift
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOTD/gpurun_out/pmc/$name" -- python "$ROOTD/bench.py" --steps 8 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-extras --no-stage-timers > "$ROOTD/gpurun_out/pmc/$name.json" 2> "$ROOTD/gpurun_out/pmc/$name.err"); }
  run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  run tcc TCC_HIT_sum TCC_MISS_sum
  run l2w TCP_TCC_WRITE_REQ_sum SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum      # L2 requests per store / load instruction (scatter: VERDICT r05 next 6)
  python tools/collect_traffic.py gpurun_out/pmc $TAG > $O/pmc_traffic_stdout.json 2> $O/collect.err
  python tools/collect_sq.py gpurun_out/pmc $TAG > $O/pmc_sq_stdout.json 2>> $O/collect.err
  cp profiles/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_sq.json $O/ 2>/dev/null
  rm -rf gpurun_out/pmc
  tail -3 $O/collect.err; ls $O ;;
tworank)
  : > $O/two_rank.log
  for ex in ${TWORANK_PLANS:-slotsum factored allreduce}; do
    echo "== $ex" >> $O/two_rank.log
    FRG_BENCH_ONE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 2 --steps 6 --warmup 3 --spinup-steps 2 --backend gloo --exchange $ex --points 400000 --no-cpu-baseline --no-extras 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^\*\*\*\|OMP_NUM" | tail -4 | cut -c1-900 >> $O/two_rank.log
  done
  cat $O/two_rank.log ;;
cmd)
  timeout ${CMD_TIMEOUT:-900} bash -c "$CMD" > $O/cmd.log 2>&1; echo "cmd rc=$?" >> $O/cmd.log; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $O/cmd.log | tail -${CMD_TAIL:-60} ;;
combine)
  # the local terms of the slot-sum exchange: phase 1, pack, the combine pass over eight views' packets (tools/combine_bench.py)
  : > $O/combine_bench.log
  IFS='|' read -ra CARGS <<< "${COMBINE_ARGS:---config c3}"
  for a in "${CARGS[@]}" ; do
    timeout 600 python tools/combine_bench.py $a 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" >> $O/combine_bench.log
  done
  cat $O/combine_bench.log | cut -c1-1500
  if [ -n "${COMBINE_PROF:-}" ]; then
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/cprof" -- python "$ROOTD/tools/combine_bench.py" ${COMBINE_PROF_ARGS:---config c3} --no-check > /dev/null 2> "$O/cprof.err")
    f=$(find $O/cprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/combine_kernel_stats.csv && head -14 $O/combine_kernel_stats.csv | cut -c1-60,140-230
    rm -rf $O/cprof
  fi ;;
combinepmc)
  # SQ / traffic / cache counters of the slot-sum exchange's kernels (pack, combine pass), separate passes
  prun() { local name="$1"; shift
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$O/$name" -- python "$ROOTD/tools/combine_bench.py" --config c3 --chunks 1 --no-check --iters 3 > /dev/null 2> "$O/$name.err")
    f=$(find $O/$name -name "*counter_collection.csv" | head -1)
    python - "$f" "$name" <<'PY'
import csv,sys,collections
f,name=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "combine" in k or "sum_rows" in k:
        acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print(name, k, {c: round(sum(v)/len(v)) for c,v in d.items()}, "launches", len(next(iter(d.values()))))
PY
    rm -rf $O/$name; }
  { prun sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
    prun sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INSTS_LDS
    prun fetch FETCH_SIZE
    prun write WRITE_SIZE
    prun tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum; } > $O/pmc_combine.log 2>&1
  cat $O/pmc_combine.log | cut -c1-400 ;;
gradab)
  # the default arithmetic's distance to float64, one A/B build (tools/build_variants.sh) at a time: $GRADAB_LIBS = names under frosting_amd/lib_ab/
  : > $O/grad_ab.log
  for v in - ${GRADAB_LIBS:-}; do
    if [ "$v" = "-" ]; then unset FROSTING_LIB; else export FROSTING_LIB="$ROOTD/frosting_amd/lib_ab/$v/libfrosting_rasterizer.so"; fi
    timeout 600 python tools/grad_ab.py ${GRADAB_ARGS:-} 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" >> $O/grad_ab.log
  done
  unset FROSTING_LIB
  cat $O/grad_ab.log ;;
ab4|ab4b)
  # alternating bench.py passes: SETTINGS="name|lib or -|bench args (commas for spaces)" CONFIGS="c3 c4 c2" (ab4b: SETTINGS2 / CONFIGS2)
  if [ $w = ab4b ]; then SET="${SETTINGS2}"; CFG="${CONFIGS2:-c3}"; else SET="${SETTINGS}"; CFG="${CONFIGS:-c3}"; fi
  : > $O/$w.log
  for rep in 1 2; do
   for setting in ${SET}; do
    name="${setting%%|*}"; rest="${setting#*|}"; lib="${rest%%|*}"; extra="${rest#*|}"; extra="${extra//,/ }"
    if [ "$lib" != "-" ]; then export FROSTING_LIB="$ROOTD/$lib"; else unset FROSTING_LIB; fi
    for cfg in ${CFG}; do
      timeout 600 python bench.py --config $cfg --steps ${STEPS:-40} --warmup 10 --no-cpu-baseline --no-extras $extra 2>> $O/$w.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %-3s %.4f ms/step | ' % ('$name', '$cfg', d['ms_per_step']) + ' '.join('%s %.3f' % (k, v) for k, v in d.get('stage_ms', {}).items()))" >> $O/$w.log
    done
   done
  done
  unset FROSTING_LIB
  cat $O/$w.log ;;
esac
done
