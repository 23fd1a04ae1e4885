#!/bin/bash
# tools/reproduce_r05.sh -- every command behind the round-5 tables in profiles/, grouped in blocks sized for one gpurun call.
# Run from the repo root on the GPU box.  Outputs go to gpurun_out/; the summaries that are kept were copied into profiles/.
set -e
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
case "${1:-help}" in
build)        # on the build box (cross-compiles without a GPU); the binaries travel with the snapshot
    make -s -C modulate_amd/csrc all && make -s -C tools tune_cycle ubench_queue_rw first_pass ubench_pcie_bidir ubench_pcie_persist ;;
pcie)         # profiles/r05_pcie_route_*.json: rocprofv3 under the routes host-resident data really takes.
              # The program stands directly behind `--` (no env / shell hop under the profiler).
    TAG=${2:-r05}
    timeout -k 10 120 tools/archive/ubench_pcie_bidir > $O/${TAG}_pcie_bidir.txt 2>&1 || true
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
feed)         # profiles/r05_pcie_persist.txt, r05_pcie_feed.txt, r05f_pcie_route_staged_*: one host-fed kernel per pageable call
    for m in 64 16 256 1024; do timeout -k 10 100 taskset -c 64-127,192-255 tools/archive/ubench_pcie_persist $m 12 >> $O/r05_persist.txt; done
    timeout -k 10 300 taskset -c 64-127,192-255 python3 tools/ab_feed.py 10 > $O/r05_ab_feed.txt
    bash tools/archive/profile_feed.sh ;;
grid)         # profiles/r05_pcie_grid.txt: workgroups of a launch across PCIe, lanes, and how a staged stream is cut
    timeout -k 10 420 python3 tools/sweep_pcie_grid.py grid > $O/r05_sweep_grid.txt
    timeout -k 10 300 python3 tools/sweep_pcie_grid.py cut > $O/r05_sweep_cut.txt ;;
wake)         # profiles/r05_first_launch.txt: the chip's first launch after an upload
    timeout -k 10 200 tools/first_pass wake 411000000 9 1500 > $O/r05_wake_411MB.txt
    timeout -k 10 200 tools/first_pass wake 4294967296 5 1500 > $O/r05_wake_4GiB.txt
    timeout -k 10 200 tools/first_pass wake 411000000 9 100 > $O/r05_wake_411MB_idle100ms.txt
    timeout -k 10 500 python3 tools/archive/first_launch_where.py 4 > $O/r05_first_launch_where.txt   # run F: events ...
    bash tools/archive/first_launch_trace.sh ;;                                                        # ... against the dispatches' own timestamps
lsp)          # profiles/r05_lsp.txt: the next chunk's loads spread over the trip, every row validated
    for n in 4294967296 411000000; do
      TUNE_ONLY="LSP|PRODUCT modgpu_cycle_queue" timeout -k 10 400 tools/tune_cycle $n 9 > $O/r05_lsp_$n.txt
    done ;;
lspcounters)  # profiles/r05_lsp_counters.json: reads in flight and fabric read latency of the LSP variants beside the product (one --pmc pass)
    D=$O/lspc; rm -rf $D; mkdir -p $D
    timeout -k 10 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum --output-format csv -d $D/pass1 -- tools/ubench_queue_rw 4294967296 200 2 > $D/pass1.log 2>&1
    timeout -k 10 120 tools/ubench_queue_rw 4294967296 200 8 > $D/rates.txt 2>&1
    python3 tools/archive/summarize_memside.py $D > $O/r05_lsp_counters.json ;;
config5)      # profiles/r05_config5_kernels.json: every kernel of configs 4 and 5 end to end (pack, save, load, extract, rebuild, save) under rocprofv3
    D=$O/r05_c5; rm -rf $D
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/bench_configs.py --sections c4,c5 --out $O/r05_config5_under_rocprofv3.json > $O/r05_config5.log 2>&1
    find $D -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/summarize_kernel_trace.py {} > $O/r05_config5_kernels.json
    find $D -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05_config5_kernel_stats.csv
    rm -rf $D ;;
soak)         # profiles/r05_soak_host.txt, r05_soak.txt, r05_every_state.txt: the final build, randomized and exhaustive
    timeout -k 10 420 python3 tools/soak_host.py 300 5 > $O/r05_soak_host.txt
    timeout -k 10 300 python3 tools/soak.py 180 17 > $O/r05_soak.txt
    timeout -k 10 300 python3 tests/_every_state_child.py > $O/r05_every_state.txt ;;
numa)         # profiles/r05_staged_numa.txt: the staged route with the caller on either socket (cpu lists of this pool's nodes: see lscpu)
    for i in 1 2 3 4; do for cpus in 64-127,192-255 0-63,128-191; do
      taskset -c $cpus modulate_amd/bin/modbench --route staged --mib 64 --reps 8 > $O/r05_numa_tmp.log
      grep "^placement\|^calls\|^staging" $O/r05_numa_tmp.log | cut -c1-130 >> $O/r05_numa.txt
    done; done ;;
bench)        # profiles/r05_bench.json, r05_pmc_summary.json (+ profiles/pmc_summary.json, which bench.py replays), r05_rocprofv3_*, r05_configs.json
    bash tools/profile.sh r05
    python3 bench.py --steps 20 --warmup 5 > $O/r05_bench.json
    python3 tools/bench_configs.py --out $O/r05_configs.json ;;
crossover)    # profiles/r05_small_call_crossover.txt: both engines per call (the table MODGPU_HOST_POLICY=fastest decides by)
    modulate_amd/bin/modbench --hostcall > $O/r05_hostcall.txt ;;
*) echo "usage: tools/reproduce_r05.sh build | pcie [tag] | feed | grid | wake | lsp | lspcounters | config5 | soak | numa | bench | crossover" ;;
esac
