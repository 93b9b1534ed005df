#!/usr/bin/env python3
"""Where inside a reverse step do the inter-kernel gaps sit?  Prints, for the first 24 steps of a kernel trace, every gap > 3 us
(position, kernel before / after) -- a gap that shows up once per 8 steps belongs to the graph launch, one per step to the step.
    python tools/gap_probe.py <kernel_trace.csv>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = "conv_first_kernel" if any("conv_first_kernel" in r["Kernel_Name"] for r in rows) else "step_prepare_kernel"
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
for n, (a, b) in enumerate(zip(starts[20:44], starts[21:45])):
    out = []
    for i in range(a, b):
        gap = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3
        if gap > 3:
            out.append(f"pos {i - a}: {gap:.1f} us ({rows[i - 1]['Kernel_Name'][:28]} -> {rows[i]['Kernel_Name'][:28]})")
    print(f"step {n}: {b - a} launches; " + ("; ".join(out) if out else "no gap > 3 us"))
