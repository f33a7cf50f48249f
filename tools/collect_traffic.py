"""Turn rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected by tools/gpu_session.sh (WHAT=pmc) in
separate --pmc runs with --kernel-trace only) into profiles/<tag>_pmc_traffic.json:
per-stage HBM-side bytes per launch, with the gfx950 correction MI355X_MICROARCH.md
prescribes (FETCH_SIZE counts a wide coalesced read at half its bytes -> doubled;
WRITE_SIZE as reported; both in KiB)."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frosting_amd import _lib
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
STAGE_OF = {"preprocess_fwd_kernel": "preprocess", "scan_kernel": "scan", "colsum_kernel": "scan", "colbase_kernel": "scan",
            "scatter_kernel": "scatter", "scatter_rows_kernel": "scatter", "reorder_kernel": "scan", "sort_tiles": "sort", "blend_fwd_kernel": "blend_fwd",
            "blend_bwd_kernel": "blend_bwd", "bwd_order_kernel": "blend_bwd", "preprocess_bwd_kernel": "preprocess_bwd",
            "big_plan_kernel": "sort", "big_chunk_sort_kernel": "sort", "big_splitters_kernel": "sort", "big_bucket_sort_kernel": "sort"}

def load(name, counter):
    files = glob.glob(os.path.join(src, name, "*", "*counter_collection.csv"))
    per_kernel = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per_kernel[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per_kernel

fetch, write = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
out = collections.defaultdict(lambda: {"fetch_kib_raw": 0.0, "write_kib_raw": 0.0})
for table, key in ((fetch, "fetch_kib_raw"), (write, "write_kib_raw")):
    launches = {}
    for kname, vals in table.items():
        stage = next((s for k, s in STAGE_OF.items() if k in kname), None)
        if stage is None:
            continue
        out[stage][key] += sum(vals) / len(vals)          # average per launch of this kernel
res = {}
for stage, d in out.items():
    res[stage] = dict(d, hbm_bytes_corrected=int((2.0 * d["fetch_kib_raw"] + d["write_kib_raw"]) * 1024))
json.dump({"build": _lib.build_fingerprint(), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py c3 (3M Gaussians, 1600x1056)",
           "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section)",
           "per_launch": res}, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
