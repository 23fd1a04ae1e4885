#!/bin/bash
# tools/reproduce_r05.sh -- every command behind the round-5 tables in profiles/, grouped in blocks sized for one gpurun call.
# Run from the repo root on the GPU box.  Outputs go to gpurun_out/; the summaries that are kept were copied into profiles/.
set -e
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
case "${1:-help}" in
build)        # on the build box (cross-compiles without a GPU); the binaries travel with the snapshot
    make -s -C modulate_amd/csrc all && make -s -C tools tune_cycle ubench_queue_rw first_pass ubench_pcie_bidir ;;
pcie)         # profiles/r05_pcie_route_*.json: rocprofv3 under the routes host-resident data really takes.
              # The program stands directly behind `--` (no env / shell hop under the profiler).
    TAG=${2:-r05}
    timeout -k 10 120 tools/ubench_pcie_bidir > $O/${TAG}_pcie_bidir.txt 2>&1 || true
    for spec in "pinned 411 10" "pinned 4096 6" "staged 16 20" "staged 64 20" "staged 256 10" "staged 1024 6"; do
      set -- $spec
      D=$O/${TAG}_pcie_$1_$2
      rm -rf $D
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- modulate_amd/bin/modbench --route $1 --mib $2 --reps $3 > $O/${TAG}_pcie_$1_$2.log 2>$O/${TAG}_pcie_$1_$2.err
      python3 tools/summarize_pcie_trace.py $O/${TAG}_pcie_$1_$2.log $D > $O/${TAG}_pcie_route_$1_$2MiB.json
      find $D -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_pcie_route_$1_$2MiB_kernel_stats.csv
      find $D -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_pcie_route_$1_$2MiB_kernel_trace.csv
      rm -rf $D
    done
    D=$O/${TAG}_pcie_config4
    rm -rf $D
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/bench_configs.py --sections c4 --out $O/${TAG}_config4_under_rocprofv3.json > $O/${TAG}_pcie_config4.log 2>&1
    find $D -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_pcie_route_config4_kernel_stats.csv
    find $D -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_pcie_route_config4_kernel_trace.csv
    rm -rf $D ;;
*) echo "usage: tools/reproduce_r05.sh build | pcie [tag]" ;;
esac
