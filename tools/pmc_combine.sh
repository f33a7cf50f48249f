cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOTD="$PWD"; O="$ROOTD/gpurun_out/${TAG:-r06pmc}"; mkdir -p $O; export TMPDIR=/tmp
run() { local name="$1"; shift
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
  rm -rf $O/$name
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
