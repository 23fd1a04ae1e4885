#!/usr/bin/env python3
"""Distil tools/archive/memside_counters.sh's rocprofv3 CSVs: per kernel (read-only / write-only / copy / product) the median per-launch
value of every counter collected, plus the derived figures DESIGN 10 quotes.  Usage: summarize_memside.py gpurun_out/memside"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def label(kernel):
    m = re.search(r"rw_kernel<\(?(?:int\))?(\d)", kernel)
    if m:
        return {"0": "read_only", "1": "write_only", "2": "copy"}[m.group(1)]
    if "modgpu_cycle_queue_kernel" in kernel:
        return "product"
    m = re.search(r"lab_cycle_queue_kernel<.*, (\d), \d>\(", kernel)  # the last template arguments are LSP, HSB (tools/cycle_kernel_lab.h)
    if m:
        return "lab_lsp_%s" % m.group(1)
    return None


def main():
    root = sys.argv[1]
    per = defaultdict(lambda: defaultdict(list))  # label -> counter -> [per-dispatch values]
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        disp = defaultdict(float)
        for r in csv.DictReader(open(f)):
            lab = label(r["Kernel_Name"])
            if lab:
                disp[(lab, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (lab, c, _), v in disp.items():
            per[lab][c].append(v)
    out = {"source": "tools/archive/memside_counters.sh (rocprofv3 --pmc, one pass per group) over tools/ubench_queue_rw", "kernels": {}}
    for lab, cs in per.items():
        out["kernels"][lab] = {c: sorted(v)[len(v) // 2] for c, v in sorted(cs.items())}
    for name in ("status.txt", "rates.txt"):
        try:
            out[name.split(".")[0]] = open(os.path.join(root, name)).read().strip().split("\n")
        except OSError:
            pass
    # derived: average fabric latency in L2 cycles (LEVEL / REQ), stalled share of L2 cycles
    for lab, k in out["kernels"].items():
        d = {}
        if k.get("TCC_EA0_RDREQ_sum"):
            d["avg_read_latency_l2_cycles"] = round(k.get("TCC_EA0_RDREQ_LEVEL_sum", 0) / k["TCC_EA0_RDREQ_sum"], 1)
        if k.get("TCC_EA0_WRREQ_sum"):
            d["avg_write_latency_l2_cycles"] = round(k.get("TCC_EA0_WRREQ_LEVEL_sum", 0) / k["TCC_EA0_WRREQ_sum"], 1)
        if k.get("TCC_EA0_RDREQ_LEVEL_sum") and k.get("TCC_CYCLE_sum"):  # TCC_CYCLE_sum adds up the 128 channels' cycles
            d["reads_in_flight_chip_wide"] = round(k["TCC_EA0_RDREQ_LEVEL_sum"] / (k["TCC_CYCLE_sum"] / 128.0))
        cyc = k.get("TCC_CYCLE_sum") or k.get("TCC_BUSY_sum")
        if cyc:
            for c in ("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_TAG_STALL_sum",
                      "TCC_TOO_MANY_EA_WRREQS_STALL_sum"):
                if c in k:
                    d[c.replace("_sum", "") + "_share_of_TCC_cycles"] = round(k[c] / cyc, 4)
        k["derived"] = d
    json.dump(out, open(os.path.join(root, "memside_counters.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
