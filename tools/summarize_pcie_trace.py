#!/usr/bin/env python3
"""tools/summarize_pcie_trace.py -- joins `modbench --route ...`'s list of launches with the rocprofv3 kernel trace of the same run.

    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- modulate_amd/bin/modbench --route staged --mib 64 > LOG
    python3 tools/summarize_pcie_trace.py LOG DIR [--ceilings CEILING_LOG] > summary.json

CEILING_LOG: the output of tools/ubench_pcie_ceiling from the SAME run on the same box.  `roofline_pcie` is priced against figures the
code under test did not produce: the link on paper (gen 5 x16: 64 GB/s per direction), the DMA engines one way and both
ways at once -- and, for reference only, what one kernel of the small shape does in place across the link.

The library's host trace names the OS thread that made every launch, rocprofv3's kernel trace names the launching thread of every
dispatch; a thread's launches and its dispatches are in the same order, which gives every dispatch its bytes, call and pipeline.
Per call (the two untimed warm-up calls are left out of the averages): kernels, bytes / duration of each, the part of the call's
kernel window [first start, last end] and of its wall clock during which at least one kernel ran, how many ran at once, and on
every queue the gaps between consecutive kernels.  "Link GB/s of a kernel" = 2 x its bytes / its duration (each byte crosses PCIe
once in each direction while the kernel runs on host memory)."""
import csv
import glob
import json
import os
import statistics as st
import sys


def read_ceilings(path):
    for line in open(path):
        if line.startswith("CEILING "):
            return json.loads(line[len("CEILING "):])
    raise SystemExit(f"{path}: no CEILING line (tools/ubench_pcie_ceiling)")


HASHES = {}


def read_log(path):
    launches, walls, head, begins, ends = [], {}, None, {}, {}
    for line in open(path):
        w = line.split()
        if not w:
            continue
        if w[0] == "hashes" and len(w) >= 5:
            HASHES.update({"kernel_source_hash": w[2], "feed_kernel_source_hash": w[4]})
        if w[0] == "route":
            head = line.strip()
        elif w[0] == "launch":
            launches.append({"tid": int(w[2]), "call": int(w[4]), "pipe": int(w[6]), "piece": int(w[8]), "bytes": int(w[10]), "t_us": float(w[12])})
        elif w[0] == "call" and w[2] == "wall_us":
            walls[int(w[1])] = float(w[3])
        elif w[0] == "callbegin":
            begins[int(w[1])] = float(w[3])
        elif w[0] == "callend":
            ends[int(w[1])] = float(w[3])
    return head, launches, walls, begins, ends


def read_trace(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "modgpu_cycle" not in r["Kernel_Name"]:
                continue
            rows.append({"tid": int(r["Thread_Id"]), "queue": int(r["Queue_Id"]), "stream": int(r.get("Stream_Id", 0) or 0), "id": int(r["Dispatch_Id"]),
                         "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"]), "kernel": r["Kernel_Name"], "grid": int(r["Grid_Size_X"]),
                         "wg": int(r["Workgroup_Size_X"])})
    rows.sort(key=lambda r: r["id"])
    return rows


def union_ns(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    log, d = sys.argv[1], sys.argv[2]
    ceil = read_ceilings(sys.argv[sys.argv.index("--ceilings") + 1]) if "--ceilings" in sys.argv else None
    link = float(ceil["peak_link"]) if ceil else 64.0  # gen 5 x16 per direction, on paper
    head, launches, walls, begins, ends = read_log(log)
    rows = read_trace(d)
    by_tid_l, by_tid_k = {}, {}
    for l in launches:
        by_tid_l.setdefault(l["tid"], []).append(l)
    for k in rows:
        by_tid_k.setdefault(k["tid"], []).append(k)
    joined, unmatched = [], 0
    for tid, ls in by_tid_l.items():
        ks = by_tid_k.get(tid, [])
        if len(ks) != len(ls):
            unmatched += abs(len(ks) - len(ls))
        for l, k in zip(ls, ks):
            joined.append({**l, **{"start": k["start"], "end": k["end"], "queue": k["queue"], "stream": k["stream"], "kernel": k["kernel"], "grid": k["grid"], "wg": k["wg"]}})
    out = {"run": head, "launches_listed": len(launches), "dispatches_traced": len(rows), "unmatched": unmatched, "link_GBps_per_direction_on_paper": link, **HASHES}
    calls = sorted({j["call"] for j in joined})
    per_call = []
    for c in calls:
        js = sorted([j for j in joined if j["call"] == c], key=lambda j: j["start"])
        if not js:
            continue
        first, last = js[0]["start"], max(j["end"] for j in js)
        busy = union_ns([(j["start"], j["end"]) for j in js])
        bytes_ = sum(j["bytes"] for j in js)
        # how many kernels run at once, time-weighted
        evs = sorted([(j["start"], 1) for j in js] + [(j["end"], -1) for j in js])
        depth, t_prev, area = 0, evs[0][0], 0
        for t, dlt in evs:
            area += depth * (t - t_prev)
            depth += dlt
            t_prev = t
        gaps = []
        for q in {j["stream"] if j["stream"] else j["queue"] for j in js}:
            qs = sorted([j for j in js if (j["stream"] if j["stream"] else j["queue"]) == q], key=lambda j: j["start"])
            gaps += [(b["start"] - a["end"]) / 1e3 for a, b in zip(qs, qs[1:])]
        wall = walls.get(c)
        per_call.append({
            "call": c, "kernels": len(js), "bytes": bytes_, "wall_us": wall,
            "kernel_window_us": round((last - first) / 1e3, 1), "busy_us": round(busy / 1e3, 1),
            "busy_frac_of_window": round(busy / (last - first), 4) if last > first else None,
            "busy_frac_of_wall": round(busy / 1e3 / wall, 4) if wall else None,
            "mean_kernels_at_once_while_busy": round(area / busy, 2) if busy else None,
            "payload_GBps_over_window": round(bytes_ / (last - first), 2) if last > first else None,
            "payload_GBps_over_wall": round(bytes_ / (wall * 1e3), 2) if wall else None,
            "first_kernel_starts_us_after_first_launch_returned": None,
            "gaps_on_a_queue_us": {"n": len(gaps), "median": round(st.median(gaps), 1) if gaps else None, "p90": round(sorted(gaps)[int(0.9 * (len(gaps) - 1))], 1) if gaps else None,
                                   "max": round(max(gaps), 1) if gaps else None, "negative_means_overlap": True},
            "kernel_kinds": sorted({j["kernel"].split("(")[0].replace("void ", "") + " grid %d x %d" % (j["grid"] // max(j["wg"], 1), j["wg"]) for j in js}),
        })
    out["per_call"] = per_call
    timed = [p for p in per_call if p["wall_us"]]
    if timed:
        out["timed_calls"] = {
            "n": len(timed),
            "median_wall_us": round(st.median(p["wall_us"] for p in timed), 1),
            "median_payload_GBps_over_wall": round(st.median(p["payload_GBps_over_wall"] for p in timed), 2),
            "median_busy_frac_of_wall": round(st.median(p["busy_frac_of_wall"] for p in timed), 4),
            "median_busy_frac_of_kernel_window": round(st.median(p["busy_frac_of_window"] for p in timed), 4),
            "median_kernels_at_once": round(st.median(p["mean_kernels_at_once_while_busy"] for p in timed), 2),
        }
    # per kernel, by piece size, timed calls only
    tcalls = {p["call"] for p in timed}
    by_size = {}
    for j in joined:
        if j["call"] in tcalls:
            by_size.setdefault(j["bytes"], []).append((j["end"] - j["start"]) / 1e3)
    out["per_kernel_by_bytes"] = [
        {"bytes": b, "n": len(v), "median_us": round(st.median(v), 1), "min_us": round(min(v), 1), "max_us": round(max(v), 1),
         "payload_GBps_at_median": round(b / st.median(v) / 1e3, 2), "link_GBps_each_way_at_median": round(b / st.median(v) / 1e3, 2),
         "frac_of_link_on_paper_at_median": round(b / st.median(v) / 1e3 / link, 3)}
        for b, v in sorted(by_size.items())]
    best = max(timed, key=lambda p: p["payload_GBps_over_wall"]) if timed else None
    if best:
        achieved = st.median(p["payload_GBps_over_wall"] for p in timed)
        r = {"bound": "pcie", "achieved": achieved, "unit": "GB/s of payload (= GB/s in each direction of the link)", "best_call": best["payload_GBps_over_wall"],
             "peak": link, "peak_is": "PCIe gen 5 x16 per direction on paper", "frac": round(achieved / link, 4), "peak_link": link, "frac_of_link": round(achieved / link, 4)}
        if ceil:
            one_way = min(ceil["dma_h2d"], ceil["dma_d2h"])
            r.update({"dma_one_way": one_way, "dma_h2d": ceil["dma_h2d"], "dma_d2h": ceil["dma_d2h"], "dma_duplex": ceil["dma_duplex_per_direction"],
                      "frac_of_dma_one_way": round(achieved / one_way, 4), "frac_of_dma_duplex": round(achieved / ceil["dma_duplex_per_direction"], 4),
                      "reference_only_one_kernel_in_place": ceil.get("product_pinned_route_in_place"),
                      "ceilings_from": "tools/ubench_pcie_ceiling, same run: DMA engines (hipMemcpyAsync, 16 MiB pieces) on %d MiB of page-locked memory" % ceil["MiB"]})
        r["dispatches_per_timed_call"] = round(st.median(p["kernels"] for p in timed), 1)
        r.update(HASHES)
        out["roofline_pcie"] = r
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
