#!/bin/bash
# tools/reproduce_r02.sh -- every command behind the round-2 tables in profiles/, in the order they were run.
# Run from the repo root on the GPU box (through gpurun, one block at a time: a block is sized to finish in minutes).
# Outputs go to gpurun_out/; the summaries that are kept were copied into profiles/ under the names given here.
set -e
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -Imodulate_amd/csrc"
build_tools() {   # on the build box (cross-compiles without a GPU); the binaries travel with the snapshot
    $HIPCC tools/tune_cycle.hip -o tools/tune_cycle
    $HIPCC tools/ubench_queue_rw.hip -o tools/ubench_queue_rw
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_d2d.hip -o tools/ubench_d2d
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/archive/ubench_latency.hip -o tools/archive/ubench_latency
}
case "${1:-help}" in
build) build_tools ;;
bench)        # profiles/r02_bench.json, r02c_* (rocprofv3 stats + PMC), r02_bench_sizes.txt
    bash tools/profile.sh r02c
    python3 bench.py > gpurun_out/r02_bench.json
    python3 tools/bench_sizes.py > gpurun_out/r02_bench_sizes.txt ;;
kernel)       # r02_tune_cycle_queue_shapes.txt / _queue_grid.txt / _lds_stage.txt / _sizes_cold.txt (rows are selected by name)
    for n in 4294967296 805306368 402653184; do timeout -k 10 400 tools/tune_cycle $n 5 > gpurun_out/r02_tune_$n.log; done
    for n in 16777216 33554432 67108864 134217728 201326592 268435456 402653184 536870912; do timeout -k 10 200 tools/tune_cycle $n 9 1 >> gpurun_out/r02_tune_sizes_cold.log; done ;;
trace)        # r02_trace_static_schedule.txt, r02_trace_queue_schedule.txt
    for n in 402653184 4294967296; do tools/tune_cycle trace $n 256 0 >> gpurun_out/r02_trace_static.log; tools/tune_cycle trace $n 256 1 >> gpurun_out/r02_trace_queue.log; done ;;
ceilings)     # r02_ubench_queue_rw.txt, r02_ubench_d2d.txt, r02_ubench_latency.txt
    for g in 200 256; do tools/ubench_queue_rw 4294967296 $g >> gpurun_out/r02_ubench_queue_rw.txt; done
    tools/ubench_d2d > gpurun_out/r02_ubench_d2d.txt; tools/ubench_d2d 1073741824 >> gpurun_out/r02_ubench_d2d.txt
    tools/archive/ubench_latency > gpurun_out/r02_ubench_latency.txt ;;
host)         # r02_sweep_pinned_routes.txt, r02_sweep_staged_routes.txt, r02_configs.json, r02_bench_hostcall_latency.txt
    python3 tools/archive/sweep_pinned.py > gpurun_out/r02_sweep_pinned.log
    python3 tools/archive/sweep_pinned.py staged > gpurun_out/r02_sweep_staged.log
    python3 tools/bench_configs.py --out gpurun_out/r02_configs.json
    python3 tools/archive/bench_hostcall.py > gpurun_out/r02_hostcall.txt ;;
*) echo "usage: tools/reproduce_r02.sh build | bench | kernel | trace | ceilings | host" ;;
esac
