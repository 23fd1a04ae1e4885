#!/bin/bash
# tools/reproduce_r04.sh -- every command behind the round-4 tables in profiles/, grouped in blocks sized for one gpurun call.
# Run from the repo root on the GPU box.  Outputs go to gpurun_out/; the summaries that are kept were copied into profiles/.
set -e
O=gpurun_out
mkdir -p $O
case "${1:-help}" in
build)        # on the build box (cross-compiles without a GPU); the binaries travel with the snapshot
    make -s -C modulate_amd/csrc all && make -s -C tools tune_cycle ubench_queue_rw first_pass ;;
bench)        # profiles/r04_bench.json, r04_* (rocprofv3 stats + PMC, replayed by bench.py as roofline.traffic), sizes, configs, batched parts
    bash tools/profile.sh r04
    python3 bench.py > $O/r04_bench.json
    python3 tools/bench_sizes.py > $O/r04_bench_sizes.txt
    python3 tools/bench_configs.py --out $O/r04_configs.json
    python3 tools/bench_batch.py > $O/r04_bench_batch.txt
    MODGPU_DEVICE_ALIAS=8 modulate_amd/bin/modbench --parts 8 --devices 0..7 --steps 5 > $O/r04_parts.txt
    modulate_amd/bin/modbench --parts 1 --steps 20 >> $O/r04_parts.txt ;;
tail)         # profiles/r04_tail.txt: where a sub-GiB launch's time goes, and the tail variants
    timeout -k 10 120 tools/tune_cycle selftest > $O/r04_selftest.txt
    for v in 0 1 2 3 4; do
      timeout -k 10 120 tools/tune_cycle trace 411000000 200 1 $v 400 0 $((v==0)) > $O/r04_trace_411MB_v$v.txt
      timeout -k 10 120 tools/tune_cycle trace 104857600 200 1 $v 400 1 $((v==0)) > $O/r04_trace_100MB_cold_v$v.txt
    done
    timeout -k 10 120 tools/tune_cycle trace 411000000 200 1 0 0 1 0 > $O/r04_trace_411MB_cold_v0.txt
    timeout -k 10 400 tools/tune_cycle 411000000 7 > $O/r04_tune_411MB.txt
    timeout -k 10 400 tools/tune_cycle 104857600 7 1 > $O/r04_tune_100MB_cold.txt
    timeout -k 10 400 tools/tune_cycle 4294967296 3 > $O/r04_tune_4GiB.txt ;;
tail2)        # profiles/r04_tail.txt, second part: the pieces in a second, cold loop (TLOOP)
    for t in 100 200 400; do
      timeout -k 10 120 tools/tune_cycle trace 411000000 200 1 5 $t 0 0 > $O/r04_trace_411MB_v5_t$t.txt
      timeout -k 10 120 tools/tune_cycle trace 411000000 200 1 6 $t 0 0 > $O/r04_trace_411MB_v6_t$t.txt
    done
    timeout -k 10 120 tools/tune_cycle trace 411000000 200 1 0 0 0 0 > $O/r04_trace_411MB_v0_again.txt
    timeout -k 10 400 tools/tune_cycle 411000000 7 > $O/r04_tune2_411MB.txt
    timeout -k 10 400 tools/tune_cycle 104857600 7 1 > $O/r04_tune2_100MB_cold.txt
    timeout -k 10 400 tools/tune_cycle 805306368 5 > $O/r04_tune2_768MiB.txt
    timeout -k 10 400 tools/tune_cycle 4294967296 3 > $O/r04_tune2_4GiB.txt ;;
tk)           # the product kernel with the ticket fetched at the start of the trip (TK): A/B against round 3's timing, the clock transient, parity
    timeout -k 10 120 tools/tune_cycle selftest > $O/r04_tk_selftest.txt
    for n in 411000000 805306368 4294967296; do timeout -k 10 400 tools/tune_cycle $n 7 > $O/r04_tk_tune_$n.txt; done
    timeout -k 10 300 tools/tune_cycle dvfs 4294967296 16 > $O/r04_tk_dvfs.txt
    python3 bench.py --no-cpu-baseline > $O/r04_tk_bench.json
    python3 tools/bench_sizes.py > $O/r04_tk_bench_sizes.txt ;;
memside)      # profiles/r04_memside_counters.json
    bash tools/archive/memside_counters.sh 4294967296 ;;
staged)       # profiles/r04_staged_midsize.txt
    modulate_amd/bin/modbench --hostcall --trace > $O/r04_hostcall_trace.txt
    python3 tools/archive/sweep_midsize_host.py > $O/r04_sweep_midsize_host.txt ;;
crossover)    # profiles/r04_small_call_crossover.txt: both engines per call, the table MODGPU_HOST_POLICY=fastest decides by
    modulate_amd/bin/modbench --hostcall > $O/r04_hostcall.txt
    MODGPU_HOST_CGROUP=0 modulate_amd/bin/modbench --hostcall > $O/r04_hostcall_nocgroup.txt
    MODGPU_HOST_SPREAD=0 modulate_amd/bin/modbench --hostcall > $O/r04_hostcall_nospread.txt ;;
parity)       # profiles/r04_every_state.txt, r04_soak.txt
    timeout -k 10 300 python3 tests/_every_state_child.py > $O/r04_every_state.txt
    timeout -k 10 600 python3 tools/soak.py 240 11 > $O/r04_soak.txt ;;
files)        # profiles/r04_file_routes.txt (incl. the I/O-only job with its source evicted from the caches: VERDICT r3 weak #8)
    modulate_amd/bin/modbench --files /dev/shm > $O/r04_files.txt ;;
extras)       # profiles/r04_first_pass.txt, r04_bench_big.txt, r04_handover.txt: round 3's tables re-taken on the final build
    for p in h2d fill; do timeout -k 10 120 tools/first_pass series 4294967296 16 $p > $O/r04_series_$p.txt; done
    timeout -k 10 400 tools/first_pass first > $O/r04_first.txt
    timeout -k 10 600 python3 tools/bench_big.py > $O/r04_bench_big.txt
    timeout -k 10 300 python3 tools/archive/handover.py > $O/r04_handover.txt ;;
*) echo "usage: tools/reproduce_r04.sh build | bench | tail | tail2 | tk | memside | staged | crossover | parity | files | extras" ;;
esac
