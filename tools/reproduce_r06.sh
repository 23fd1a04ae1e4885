#!/bin/bash
# tools/reproduce_r06.sh <block> -- the exact commands behind every round-6 table in profiles/ (run on the GPU box through gpurun, from
# the repo root; every block writes under gpurun_out/, the summaries judged are then copied to profiles/).  Nothing may stand between
# rocprofv3's `--` and the program.
#   ceiling     tools/ubench_pcie_ceiling: DMA one way / both ways, a kernel across PCIe read-only / write-only / copy / in place  -> r06_pcie_ceiling.txt
#   fileroutes  tools/ab_file_feed.py: file -> memory, host-fed kernel against a launch per chunk, interleaved                      -> r06_file_routes.txt
#   pcie        modbench --route <kind> under rocprofv3 --kernel-trace, joined with the library's launch list, priced against `ceiling` -> r06_pcie_route_*.json
#   bench       python3 bench.py (the driver's line) and tools/profile.sh r06 (rocprofv3 stats + PMC)                               -> r06_bench.json, r06_*
#   gpus2       python3 bench.py --gpus 2 by itself on the one GPU                                                                  -> r06_bench_gpus2_by_itself.json
#   configs     BASELINE configs 1, 4, 5 end to end + the host-buffer rates with roofline_pcie                                      -> r06_configs.json
set -e
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
mkdir -p $O
case "$1" in
ceiling)
  tools/ubench_pcie_ceiling 1024 5 > $O/r06_pcie_ceiling.txt 2>&1
  tail -20 $O/r06_pcie_ceiling.txt ;;
fileroutes)
  python3 tools/ab_file_feed.py 5 64,392,4096 > $O/r06_file_routes.txt 2>&1
  cat $O/r06_file_routes.txt ;;
pcie)
  [ -s $O/r06_pcie_ceiling.txt ] || tools/ubench_pcie_ceiling 1024 5 > $O/r06_pcie_ceiling.txt 2>&1
  for spec in "pinned 1024 6" "staged 16 20" "staged 64 20" "staged 256 10" "staged 1024 6" "file_pageable 64 20" "file_pageable 392 10" "file_pinned 64 20" "file_pinned 392 10" "file_pinned 1024 6"; do
    set -- $spec
    D=$O/r06_pcie_$1_$2
    rm -rf $D
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- modulate_amd/bin/modbench --route $1 --mib $2 --reps $3 --socket near > $O/r06_pcie_$1_$2.log 2> $O/r06_pcie_$1_$2.err
    python3 tools/summarize_pcie_trace.py $O/r06_pcie_$1_$2.log $D --ceilings $O/r06_pcie_ceiling.txt > $O/r06_pcie_route_$1_$2MiB.json
    find $D -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r06_pcie_route_$1_$2MiB_kernel_stats.csv
    rm -rf $D
    python3 -c "import json,sys; d=json.load(open('$O/r06_pcie_route_$1_$2MiB.json')); r=d['roofline_pcie']; print('$1 $2 MiB', r['achieved'], 'GB/s  frac_of_link', r['frac_of_link'], ' of dma one way', r.get('frac_of_dma_one_way'), ' dispatches per call', r['dispatches_per_timed_call'])"
  done ;;
bench)
  python3 bench.py --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/r06_bench.err
  bash tools/profile.sh r06 > $O/r06_profile.log 2>&1
  tail -5 $O/r06_profile.log ;;
gpus2)
  python3 bench.py --gpus 2 --force-device 0 --steps 3 --warmup 1 --part-bytes 335544320 > $O/r06_bench_gpus2_by_itself.json ;;
configs)
  python3 tools/bench_configs.py --out $O/r06_configs.json > $O/r06_configs.log 2>&1
  tail -3 $O/r06_configs.log ;;
*) echo "blocks: ceiling fileroutes pcie bench gpus2 configs"; exit 1 ;;
esac
