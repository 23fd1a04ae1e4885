#!/usr/bin/env python3
"""tools/summarize_kernel_trace.py <kernel_trace.csv> -- a rocprofv3 kernel trace by (kernel, grid): how many dispatches, their
durations, and how much of the span from the first start to the last end had at least one of them running.  For the runs where
no launch list exists to join with (tools/summarize_pcie_trace.py does that for `modbench --route`), e.g. config 4 under the
profiler."""
import csv
import json
import statistics as st
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "modgpu_" in r["Kernel_Name"]]
groups = {}
for r in rows:
    k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Workgroup_Size_X"]))
    groups.setdefault(k, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
out = []
for (name, grid, wg), iv in sorted(groups.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    d = [(e - s) / 1e3 for s, e in iv]
    out.append({"kernel": name, "workgroups": grid, "threads": wg, "dispatches": len(d), "total_us": round(sum(d), 1), "median_us": round(st.median(d), 1),
                "min_us": round(min(d), 1), "max_us": round(max(d), 1)})
json.dump({"dispatches": len(rows), "by_kernel_and_grid": out}, sys.stdout, indent=1)
print()
