#!/usr/bin/env bash
# one lane-packing variant of the forward blend MEASURED on the GPU -- the A/B build FRG_AB_HALVES (the
# quadrant's two 8x4 halves walk their own culled lists side by side, tools/build_variants.sh halves=-DFRG_AB_HALVES
# u1=-DFRG_FWD_UNROLL=1) against the product build: parity tests through the ctypes binding, alternating bench passes, and the
# forward blend's vector-instruction counters.  PARTS selects (default: all).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTD=$PWD; O=$ROOTD/gpurun_out/r05p; mkdir -p $O
export TMPDIR=/tmp
H=$ROOTD/frosting_amd/lib_ab/halves/libfrosting_rasterizer.so
U1=$ROOTD/frosting_amd/lib_ab/u1/libfrosting_rasterizer.so
for part in ${PARTS:-tests ab pmc}; do
case $part in
tests)
  FROSTING_LIB=$H timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "(bit_exact and ctypes) or c_oracle or ragged or deep_walks and ctypes or long_lists and ctypes" > $O/pytest_halves.log 2>&1; tail -3 $O/pytest_halves.log ;;
ab)
  TAG=r05p WHAT="ab4" SETTINGS="cur|-| u1|frosting_amd/lib_ab/u1/libfrosting_rasterizer.so| halves|frosting_amd/lib_ab/halves/libfrosting_rasterizer.so|" CONFIGS="c3 c4 c2" STEPS=40 bash tools/gpu_round5.sh | tail -20 ;;
pmc)
  : > $O/pmc_blend_fwd.log
  for v in cur u1 halves; do
    if [ $v = cur ]; then unset FROSTING_LIB; elif [ $v = u1 ]; then export FROSTING_LIB=$U1; else export FROSTING_LIB=$H; fi
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d "$O/pmc_$v" -- python "$ROOTD/bench.py" --steps 8 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-extras --no-stage-timers > "$O/pmc_$v.json" 2> "$O/pmc_$v.err")
    python - "$O/pmc_$v" $v >> $O/pmc_blend_fwd.log <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "blend_fwd_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], "blend_fwd_kernel per launch:", {c:int(sum(v)/len(v)) for c,v in acc.items()})
PY
    rm -rf "$O/pmc_$v"
  done
  unset FROSTING_LIB
  cat $O/pmc_blend_fwd.log ;;
esac
done
