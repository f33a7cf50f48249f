"""Average per launch of every SQ / TCC counter of the PMC passes (tools/gpu_session.sh (WHAT=pmc): sq1, sq2, tcc) per kernel ->
profiles/<tag>_pmc_sq.json.  Units as rocprofv3 reports them (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles, MI355X_MICROARCH.md)."""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frosting_amd import _lib
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("sq1", "sq2", "tcc", "l2w"):
    for f in glob.glob(os.path.join(src, name, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "frg::" not in k:
                continue
            k = re.sub(r"\(.*", "", k).replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: int(sum(v) / len(v)) for c, v in cs.items()} for k, cs in acc.items()}
json.dump({"build": _lib.build_fingerprint(), "source": "rocprofv3 --kernel-trace --pmc <counters> (separate passes, tools/gpu_session.sh (WHAT=pmc)), bench.py c3; average per launch",
           "kernels": out}, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_sq.json"), "w"), indent=1)
print(json.dumps({k: v.get("SQ_INSTS_VALU") for k, v in out.items()}, indent=1))
