"""Timeline of one step from a rocprofv3 --kernel-trace CSV: every kernel of the LAST complete step (from the last
preprocess_fwd_kernel to the end of the step's last kernel) with start offset, duration and the idle gap in front of it.
usage: python tools/trace_gaps.py <kernel_trace.csv> [first-kernel-substring [must-contain-substring]]
(with the third argument: the last complete step that holds a kernel of that name, e.g. combine_tile for an exchanging step)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "preprocess_fwd_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
if len(starts) < 3:
    raise SystemExit("fewer than three steps in the trace")
a, b = starts[-2], starts[-1]
if len(sys.argv) > 3:
    for k in range(len(starts) - 2, 0, -1):
        if any(sys.argv[3] in r["Kernel_Name"] for r in rows[starts[k]:starts[k + 1]]):
            a, b = starts[k], starts[k + 1]
            break
    else:
        raise SystemExit(f"no step holds a kernel named *{sys.argv[3]}*")
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
end_prev = t0
print(f"{'start us':>9s} {'dur us':>8s} {'gap us':>7s}  kernel")
busy = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("frg::", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - end_prev) / 1e3:7.1f}  {name}")
    end_prev = max(end_prev, e)
    busy += e - s
print(f"step: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us from first kernel to the next step's first kernel; sum of kernel durations {busy / 1e3:.1f} us")
