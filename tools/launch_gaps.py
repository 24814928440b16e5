#!/usr/bin/env python3
"""What a launch boundary costs: from a rocprofv3 --kernel-trace CSV, the iteration kernels' durations (End - Start of a dispatch)
and the gaps between consecutive ones on the stream (Start of the next - End of this one).  A step of the block loop is one launch:
kernel + gap is what `ms_per_step` pays.      launch_gaps.py <..._kernel_trace.csv> [name substring = iteration_kernel]"""
import csv, statistics, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
pat = sys.argv[2] if len(sys.argv) > 2 else "iteration_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap, names = [], [], {}
for a, b in zip(rows, rows[1:]):
    if pat in a["Kernel_Name"]:
        dur.append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
        names[a["Kernel_Name"].split("(")[0][-60:]] = names.get(a["Kernel_Name"].split("(")[0][-60:], 0) + 1
        if pat in b["Kernel_Name"]:
            gap.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
if not dur:
    sys.exit("no such kernel in the trace")
q = lambda v, p: sorted(v)[min(len(v) - 1, int(p * len(v)))]
print(f"{len(dur)} launches of *{pat}*: duration  p10 {q(dur, .1):.1f}  p50 {statistics.median(dur):.1f}  p90 {q(dur, .9):.1f} us;  "
      f"gap to the next launch ({len(gap)})  p10 {q(gap, .1):.2f}  p50 {statistics.median(gap):.2f}  p90 {q(gap, .9):.2f} us;  "
      f"gap / (duration + gap) = {statistics.median(gap) / (statistics.median(dur) + statistics.median(gap)):.3f}")
for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:4]:
    print(f"   {v:5d} x {k}")
