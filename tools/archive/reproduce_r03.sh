#!/bin/bash
# tools/reproduce_r03.sh -- every command behind the round-3 tables in profiles/, grouped in blocks sized for one gpurun call.
# Run from the repo root on the GPU box.  Outputs go to gpurun_out/; the summaries that are kept were copied into profiles/.
set -e
case "${1:-help}" in
build)        # on the build box (cross-compiles without a GPU); the binaries travel with the snapshot
    make -s -C modulate_amd/csrc all && make -s -C tools tune_cycle first_pass ;;
bench)        # profiles/r03_bench.json, r03_* (rocprofv3 stats + PMC, replayed by bench.py as roofline.traffic)
    bash tools/profile.sh r03
    python3 bench.py > gpurun_out/r03_bench.json ;;
first_pass)   # profiles/r03_first_pass.txt, r03_tune_dvfs.txt
    for p in h2d fill; do timeout -k 10 120 tools/first_pass series 4294967296 16 $p > gpurun_out/r03_series_$p.txt; done
    timeout -k 10 120 tools/first_pass series 4294967296 16 h2d 200 > gpurun_out/r03_series_h2d_idle200.txt
    timeout -k 10 400 tools/first_pass first > gpurun_out/r03_first.txt
    timeout -k 10 300 tools/tune_cycle dvfs 4294967296 16 > gpurun_out/r03_tune_dvfs.txt ;;
kernel)       # profiles/r03_tune_cycle_alg.txt
    for n in 4294967296 805306368 411000000; do timeout -k 10 400 tools/tune_cycle $n 5 > gpurun_out/r03_tune_$n.txt; done
    python3 tools/bench_sizes.py > gpurun_out/r03_bench_sizes.txt ;;
host)         # profiles/r03_small_call_crossover.txt, r03_file_routes.txt, r03_parts_one_process.txt, r03_configs.json
    modulate_amd/bin/modbench --hostcall > gpurun_out/r03_hostcall.txt
    modulate_amd/bin/modbench --files /dev/shm > gpurun_out/r03_files.txt
    modulate_amd/bin/modbench --alloc > gpurun_out/r03_alloc.txt
    MODGPU_DEVICE_ALIAS=8 modulate_amd/bin/modbench --parts 8 --steps 5 > gpurun_out/r03_parts.txt
    modulate_amd/bin/modbench --parts 1 --steps 20 >> gpurun_out/r03_parts.txt
    modulate_amd/bin/modbench 4294967296 20 3 >> gpurun_out/r03_parts.txt
    python3 tools/bench_configs.py --out gpurun_out/r03_configs.json
    python3 tools/bench_batch.py > gpurun_out/r03_bench_batch.txt        # profiles/r03_parts_batched.txt
    modulate_amd/bin/modbench --parts 8 --devices 0 --part-bytes 411000000 --steps 40 --warmup 5 >> gpurun_out/r03_bench_batch.txt ;;
parity)       # profiles/r03_every_state.txt, r03_soak.txt, r03_numa.txt
    timeout -k 10 300 python3 tests/_every_state_child.py > gpurun_out/r03_every_state.txt
    timeout -k 10 600 python3 tools/soak.py 240 7 > gpurun_out/r03_soak.txt
    modulate_amd/bin/modbench --numa > gpurun_out/r03_numa.txt ;;
*) echo "usage: tools/reproduce_r03.sh build | bench | first_pass | kernel | host | parity" ;;
esac
