"""modulate_amd -- MI355X (gfx950) implementation of Modulate's cipher hot path.

The product is the C-ABI shared library ``libmodgpu.so`` (include/modgpu.h), built from
``modulate_amd/csrc`` with hipcc, plus the C++ host mirror of the reference surface
(``CEncryptionCycler``, the ``CArk`` buffer path) in ``libmodulate_host.so``.  This Python
package is only a ctypes binding over those libraries for the test-suite and ``bench.py``;
there is no Python implementation of the cipher in it, and importing the bindings fails loudly if
the HIP library has not been built.  (The library's own host loop, for machines without a GPU, lives
in libmodgpu.so itself: modgpu_cycle_scalar_host / modgpu_cycle_auto_host, include/modgpu.h.)
"""
from .capi import (  # noqa: F401
    ModGpuError, lib, lib_path, device_count, cycle_host, cycle_device, hdr_decrypt_host,
    hdr_encrypt_host, cycle_parts_host, cycle_parts_device, cycle_batch_device, cycle_host_split, debug_set_batch, cycle_file, cycle_file_to_host, cycle_host_to_file, DeviceBuffer, time_cycle_device, state_at, jump_table,
    KEY_PS3, KEY_PS4, MAGIC_PS3, MAGIC_PS4, as_int32, EXPORTS, TESTING_EXPORTS, cycle_scalar_host, cycle_auto_host,
    path_stats, gpu_required, last_launch, debug_set_launch, debug_set_pinned_mode, debug_set_staged_mode, kernel_source_hash, feed_kernel_source_hash, prepare, host_tunables, debug_inject_failures, debug_inject_failure_at, debug_injection_armed, debug_hold_slots, debug_forbid_worker_threads, debug_set_pcie_grid, debug_set_gpu_node, debug_set_host_tunable, HOST_TUNABLES, STAGE_FILL, STAGE_LAUNCH, STAGE_SYNC, STAGE_DRAIN, STAGE_AFTER_DRAIN, STAGE_STALL, INJECT_PIECE_LAST, INJECT_PIECE_MIDDLE, PinnedBuffer, host_register, host_unregister,
    DEBUG_EXPORTS, FLAVOURS, testing_flavour, use_testing_flavour, active_flavour, debug_set_queue_ring, debug_set_helpers, queue_stats, testing_hooks, min_gpu_bytes,
    host_loop_isa, cycle_scalar_host_isa, device_numa_node, numa_probe, host_policy, host_policy_engine, host_trace, host_trace_read, host_pool_stats, host_chunking, HOST_TRACE_KINDS,
)
