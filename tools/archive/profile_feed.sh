#!/bin/bash
# tools/archive/profile_feed.sh -- the pageable host route with the host-fed kernel under rocprofv3 (what `reproduce_r05.sh pcie` does for the staged
# rows, after the route changed): modbench --route staged under --kernel-trace, every dispatch joined with the library's launch list.
# The calling thread is put on the GPU's socket (and once, at 64 MiB, on the other) by modbench itself: nothing may stand between `--` and the program.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; TAG=r05f
for spec in "staged 16 20 near" "staged 64 20 near" "staged 256 10 near" "staged 1024 6 near" "staged 64 20 far"; do
  set -- $spec
  S=$4; [ "$S" = far ] && TAG=r05f_far || TAG=r05f
  D=$O/${TAG}_pcie_$1_$2
  rm -rf $D
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- modulate_amd/bin/modbench --route $1 --mib $2 --reps $3 --socket $S > $O/${TAG}_pcie_$1_$2.log 2>$O/${TAG}_pcie_$1_$2.err || exit 1
  python3 tools/summarize_pcie_trace.py $O/${TAG}_pcie_$1_$2.log $D > $O/${TAG}_pcie_route_$1_$2MiB.json
  find $D -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_pcie_route_$1_$2MiB_kernel_stats.csv
  rm -rf $D
done
