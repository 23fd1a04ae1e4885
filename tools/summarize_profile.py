#!/usr/bin/env python3
"""Distil rocprofv3 CSV output (tools/profile.sh) into small, committable summaries:
   <out>/<tag>_kernel_stats.csv   per-kernel count / total / average duration
   <out>/<tag>_pmc_summary.json   FETCH_SIZE / WRITE_SIZE per launch of the cycle kernel,
                                  corrected as MI355X_MICROARCH.md (HBM) prescribes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pat):
    return sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))


def valu_utilisation(sq, cus=256, xcds=8):
    """VALU utilisation beside the HBM figure.  counter() sums a dispatch's rows, so GRBM_GUI_ACTIVE is the sum
    over the XCDs.  rocprofv3's own VALUBusy (counter_defs.yaml, gfx94x formula) prices every VALU instruction at one
    quad-cycle (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU on gfx950); this kernel's mix is two 8-cycle v_mad_u64_u32 per
    4-cycle instruction -- 6.6 cycles per instruction measured where the SIMDs are saturated (profiles/r03_first_pass.txt)."""
    if not all(k in sq for k in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")):
        return {}
    cycles = sq["GRBM_GUI_ACTIVE"] / xcds
    simds = 4 * cus
    return {"valu": {"gpu_cycles_per_launch": round(cycles),
                     "VALUBusy_pct_rocprofv3_formula": round(100 * sq["SQ_ACTIVE_INST_VALU"] / cus / cycles, 1),
                     "issue_cycles_per_instruction_measured_saturated": 6.6,
                     "valu_issue_pct_of_all_simds": round(100 * sq["SQ_INSTS_VALU"] * 6.6 / simds / cycles, 1),
                     "valu_issue_pct_of_the_800_simds_of_the_main_workgroups": round(100 * sq["SQ_INSTS_VALU"] * 6.6 / 800 / cycles, 1)}}


def main():
    root, tag = sys.argv[1], sys.argv[2]
    part_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 32
    out = {"tag": tag, "part_bytes": part_bytes,
           "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py ; rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE "
                      f"(separate passes) -- python3 bench.py --no-cpu-baseline  (tools/profile.sh {tag})"}
    # identity of the device code these figures belong to: bench.py replays `traffic` only for the same hash
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import modulate_amd as M
        out["kernel_source_hash"] = M.kernel_source_hash()
    except Exception as e:  # noqa: BLE001
        out["kernel_source_hash"] = None
        out["kernel_source_hash_error"] = str(e)
    # ---- kernel trace
    per = defaultdict(list)
    for f in find(os.path.join(root, "trace"), "*kernel_trace.csv"):
        for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
            per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        rows.append({"kernel": k, "calls": len(v), "total_ns": sum(v), "avg_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v)})
    with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns"])
        w.writeheader()
        w.writerows(rows)
    for r in rows[:5]:
        print(f"{r['calls']:6d} x avg {r['avg_ns']/1e3:10.1f} us  {r['kernel'][:90]}")
    cyc = [r for r in rows if "modgpu_cycle_" in r["kernel"]]  # rows are sorted by total time: [0] is the dominant kernel
    if cyc:
        out["cycle_kernel"] = cyc[0]["kernel"]
        out["avg_launch_ns_traced"] = cyc[0]["avg_ns"]
        # bench.py says which launches of this kernel its HIP events bracket (roofline.timed_launches: the first-pass
        # preamble and the warm-up come before, two check launches after); rocprofv3's --stats table averages all of them.
        d = [x for x in per[cyc[0]["kernel"]] if x > 20000]  # (without the tiny launches modgpu_alloc's device preparation makes: < 10 us)
        lo = hi = None
        for f in find(root, "bench_trace.log"):
            for line in open(f):
                if line.startswith("{"):
                    try:
                        lo, hi = json.loads(line)["roofline"]["timed_launches"]
                    except (ValueError, KeyError):
                        pass
        if lo is not None and hi <= len(d):
            timed = d[lo:hi]
            out["timed_launches"] = [lo, hi]
            out["launches_traced"] = len(d)
            out["avg_launch_ns_traced_timed_region"] = sum(timed) / len(timed)
            out["median_launch_ns_traced"] = sorted(d)[len(d) // 2]
            out["first_launches_ns_traced"] = d[:16]
    # ---- counters
    def counter(dirname, name):
        vals = defaultdict(float)
        for f in find(os.path.join(root, dirname), "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                # the dominant kernel only: since round 5 a run also holds dozens of EMPTY launches of the small shape (modgpu_h2d
                # wakes the shader engines when a copy starts) and the device preparation's two tiny work-queue launches
                if r["Kernel_Name"] == out.get("cycle_kernel", r["Kernel_Name"]) and "modgpu_cycle_" in r["Kernel_Name"] and r["Counter_Name"] == name:
                    vals[(r["Dispatch_Id"])] += float(r["Counter_Value"])
        return list(vals.values())
    fetch = counter("pmc_fetch", "FETCH_SIZE")
    write = counter("pmc_write", "WRITE_SIZE")
    if fetch and write:
        # keep the 4 GiB launches only (largest values), take the median
        fetch.sort(); write.sort()
        f_kb, w_kb = fetch[len(fetch) // 2], write[len(write) // 2]
        # MI355X_MICROARCH.md HBM: counters are in KiB... FETCH_SIZE reads 1/2 of a wide coalesced
        # 16 B/lane stream on gfx950 -> double it; WRITE_SIZE is exact for 16 B/lane stores.
        out.update({"FETCH_SIZE_kb_median": f_kb, "WRITE_SIZE_kb_median": w_kb,
                    "fetch_bytes_corrected": 2 * f_kb * 1024, "write_bytes": w_kb * 1024,
                    "hbm_bytes_per_launch": 2 * f_kb * 1024 + w_kb * 1024,
                    "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE as is; KiB units"})
    sq = {}
    for name in ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
        v = counter("pmc_sq", name)
        if v:
            v.sort()
            sq[name] = v[len(v) // 2]
    if sq:
        out["sq_median_per_launch"] = sq
        out.update(valu_utilisation(sq))
    with open(os.path.join(root, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
