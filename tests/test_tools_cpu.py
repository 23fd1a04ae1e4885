"""CPU checks of the measurement tooling that turns profiler output into the figures DESIGN.md quotes (no GPU, no profiler needed:
synthetic inputs in the formats `bin/modbench --route` and `rocprofv3 --kernel-trace --output-format csv` write)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEADER = ('"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp",'
          '"LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size_X","Workgroup_Size_Y","Workgroup_Size_Z","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"')


def row(queue, tid, disp, start, end, grid=32 * 256):
    return f'"KERNEL_DISPATCH","Agent 2",{queue},{queue},{tid},{disp},11,"void modgpu_cycle_kernel<1, 256, 1, false>(CycleArgs)",{disp},{start},{end},0,0,43,0,48,256,1,1,{grid},1,1'


def test_pcie_trace_join_by_thread_and_order(tmp_path):
    """tools/summarize_pcie_trace.py: a thread's launches (the library's host trace) and its dispatches (the profiler's kernel trace) are in
    the same order; the join gives every kernel its bytes and call, and from that bytes / duration, the busy share of a call's wall
    clock, how many kernels ran at once and the gaps on a queue."""
    log = tmp_path / "route.log"
    # two calls (call 0 is the untimed warm-up: no wall time), two pipeline threads, pieces of 1 and 4 MiB
    log.write_text("\n".join([
        "route staged  bytes 10485760  reps 1",
        "hashes kernel ddcc feed ffee",
        "call 1 wall_us 400.0",
        "callbegin 0 t_us 0.0", "launch tid 101 call 0 pipe 0 piece 0 bytes 1048576 t_us 10.0", "launch tid 102 call 0 pipe 1 piece 1 bytes 4194304 t_us 12.0", "callend 0 t_us 300.0",
        "callbegin 1 t_us 1000.0", "launch tid 101 call 1 pipe 0 piece 0 bytes 1048576 t_us 1010.0", "launch tid 102 call 1 pipe 1 piece 1 bytes 4194304 t_us 1012.0",
        "launch tid 101 call 1 pipe 0 piece 2 bytes 4194304 t_us 1100.0", "callend 1 t_us 1400.0", ""]))
    d = tmp_path / "prof" / "host"
    d.mkdir(parents=True)
    us = 1000
    (d / "1_kernel_trace.csv").write_text("\n".join([HEADER,
        '"KERNEL_DISPATCH","Agent 2",1,1,101,1,9,"__amd_rocclr_fillBufferAligned",1,1,2,0,0,8,0,48,256,1,1,1024,1,1',
        row(1, 101, 2, 5000 * us, 5040 * us), row(2, 102, 3, 5010 * us, 5110 * us),                      # call 0
        row(1, 101, 4, 9000 * us, 9050 * us), row(2, 102, 5, 9020 * us, 9220 * us), row(1, 101, 6, 9100 * us, 9300 * us), ""]))  # call 1
    # what tools/ubench_pcie_ceiling prints last, from the same run: the figures the roofline is priced against
    ceil = tmp_path / "ceiling.txt"
    ceil.write_text('== table\nCEILING {"MiB": 1024, "peak_link": 64.0, "dma_h2d": 55.0, "dma_d2h": 56.0, "dma_duplex_per_direction": 47.0, '
                    '"product_pinned_route_in_place": 50.4, "kernel_source_hash": "aa", "feed_kernel_source_hash": "bb"}\n')
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pcie_trace.py"), str(log), str(tmp_path / "prof"), "--ceilings", str(ceil)],
                       capture_output=True, text=True, check=True)
    out = json.loads(r.stdout)
    assert out["launches_listed"] == 5 and out["dispatches_traced"] == 5 and out["unmatched"] == 0
    c1 = [c for c in out["per_call"] if c["call"] == 1][0]
    assert c1["kernels"] == 3 and c1["bytes"] == 9 << 20 and c1["wall_us"] == 400.0
    assert c1["kernel_window_us"] == 300.0 and c1["busy_us"] == 300.0  # [9000, 9300] us, covered without a hole
    assert abs(c1["busy_frac_of_wall"] - 0.75) < 1e-9
    assert abs(c1["mean_kernels_at_once_while_busy"] - (50 + 200 + 200) / 300) < 0.01
    assert c1["gaps_on_a_queue_us"]["n"] == 1 and c1["gaps_on_a_queue_us"]["median"] == 50.0  # queue 1: 9050 -> 9100
    by = {k["bytes"]: k for k in out["per_kernel_by_bytes"]}   # timed calls only: call 1
    assert by[1 << 20]["n"] == 1 and by[1 << 20]["median_us"] == 50.0 and by[4 << 20]["n"] == 2 and by[4 << 20]["median_us"] == 200.0
    assert abs(by[4 << 20]["payload_GBps_at_median"] - (4 << 20) / 200e-6 / 1e9) < 0.01
    # the peak is the link on paper, with the DMA engines' figures of the same run beside it -- nothing the library's own kernel produced
    rf = out["roofline_pcie"]
    ach = (9 << 20) / 400e-6 / 1e9
    assert rf["bound"] == "pcie" and rf["peak"] == rf["peak_link"] == 64.0 and abs(rf["achieved"] - ach) < 0.01
    assert abs(rf["frac_of_link"] - ach / 64.0) < 1e-3 and rf["dma_one_way"] == 55.0 and rf["dma_duplex"] == 47.0
    assert abs(rf["frac_of_dma_one_way"] - ach / 55.0) < 1e-3 and abs(rf["frac_of_dma_duplex"] - ach / 47.0) < 1e-3
    assert rf["reference_only_one_kernel_in_place"] == 50.4 and rf["dispatches_per_timed_call"] == 3
    assert out["feed_kernel_source_hash"] == rf["feed_kernel_source_hash"] == "ffee" and out["kernel_source_hash"] == "ddcc"


def test_kernel_trace_summary_by_kernel_and_grid(tmp_path):
    p = tmp_path / "k_kernel_trace.csv"
    p.write_text("\n".join([HEADER, row(1, 7, 1, 0, 8_000_000, grid=256 * 256), row(1, 7, 2, 9_000_000, 17_200_000, grid=256 * 256), row(2, 8, 3, 100, 5600, grid=2048 * 256), ""]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_kernel_trace.py"), str(p)], capture_output=True, text=True, check=True)
    out = json.loads(r.stdout)
    assert out["dispatches"] == 3
    big = out["by_kernel_and_grid"][0]
    assert big["workgroups"] == 256 and big["dispatches"] == 2 and big["median_us"] == 8100.0 and big["total_us"] == 16200.0
    assert out["by_kernel_and_grid"][1]["workgroups"] == 2048 and out["by_kernel_and_grid"][1]["median_us"] == 5.5
