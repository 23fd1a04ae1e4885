#!/bin/bash
# tools/archive/first_launch_trace.sh -- run F3 of profiles/r05_first_launch.txt: children of tools/archive/first_launch_where.py under rocprofv3 --kernel-trace,
# HIP events (first, next, mean of 8, upload s) against the dispatches' own begin / end timestamps.  Appends to gpurun_out/r05_first_launch_where4.txt.
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
for c in alloc4G_up4G_head alloc411_up411 alloc4G_up4G_head_raw alloc4G_up4G_head; do
  D=$O/tw_$c; rm -rf $D
  echo "== $c" >> $O/r05_first_launch_where4.txt
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 tools/archive/first_launch_where.py --child $c >> $O/r05_first_launch_where4.txt 2>$O/tw_err.txt
  f=$(find $D -name "*kernel_trace.csv" | head -1)
  python3 - "$f" >> $O/r05_first_launch_where4.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q = [r for r in rows if "queue" in r["Kernel_Name"] and int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]) >= 200 * 1024]
print("   queue-kernel dispatches with a full grid, by the dispatch's own timestamps (ms):", " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.4f}" for r in q[:10]))
PY
  rm -rf $D
done
cat $O/r05_first_launch_where4.txt
