// modgpu_internal.h -- shared by the host-side translation units of libmodgpu.so
// (modgpu_capi.cpp: devices, launch planning, the ABI; host_stream.cpp: host-buffer and file
// endpoints).  Not installed; the public surface is include/modgpu.h.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <string>

#include "../../include/modgpu.h"

namespace modgpu {

// ---- errors (text per calling thread) -------------------------------------------------------
extern thread_local std::string t_err;
int fail(int code, const char *what);
int fail(int code, const std::string &what);
int fail_hip(hipError_t e, const char *where);
int fail_io(const char *what); // MODGPU_ERR_IO, text = what + strerror(errno)

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return ::modgpu::fail_hip(e_, #expr);                                \
    } while (0)

// Body of an extern "C" entry point: nothing C++ may leave through the C boundary.
template <typename F> int guarded(F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(MODGPU_ERR_INVALID, "out of host memory");
    } catch (const std::exception &e) {
        return fail(MODGPU_ERR_INVALID, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(MODGPU_ERR_INVALID, "internal error");
    }
}

// ---- devices ----------------------------------------------------------------------------------
// `device` arguments of the ABI are LOGICAL indices: normally the HIP ordinal; under
// MODGPU_DEVICE_ALIAS=N there are N of them and logical d runs on HIP device d mod <visible>.
constexpr int kMaxDevices = 64;
int physical_count();
int logical_count();
int physical_of(int logical);
// Makes the device current for the calling thread (HIP's current device is per thread).
// device < 0 keeps the thread's current device.
int select_device(int device);
// Same, and returns the logical index in *out (the current HIP ordinal when device < 0).
int resolve_device(int device, int *out);
// select_device for the length of a scope: an entry point called with device >= 0 works on that GPU and puts the
// calling thread's current HIP device back when it returns (a torch process must not find its device switched).
class DeviceScope {
public:
    explicit DeviceScope(int device, bool always_save = false);
    ~DeviceScope();
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
    int rc = MODGPU_OK; // select_device's status

private:
    int prev_ = -1;
};

// NUMA node of the GPU behind a logical device (-1 unknown / MODGPU_NUMA=0); and: restrict the calling (worker)
// thread to that node's CPUs.  Both best effort.
int device_numa_node(int logical);
void run_near_device(int logical);

// ---- the launch (device already current) ----------------------------------------------------
// over_pcie: dev_buf is page-locked HOST memory the kernel reaches across PCIe -- planned with the shape
// and grid that saturate the link instead of the ones that saturate HBM.
int cycle_device_impl(void *dev_buf, uint64_t n, int32_t key, uint64_t stream_off, hipStream_t stream,
                      bool over_pcie = false);

// ---- which engine ran -------------------------------------------------------------------------
struct Stats {
    std::atomic<uint64_t> gpu_calls{0}, gpu_bytes{0}, gpu_launches{0}, scalar_calls{0}, scalar_bytes{0},
        staged_bytes{0}, direct_bytes{0}, auto_fallbacks{0}, auto_small{0}, auto_policy_host{0}, midcall_rescues{0}, midcall_rescued_bytes{0};
};
extern Stats g_stats;
bool gpu_required(); // MODGPU_REQUIRE_GPU=1 at load

// ---- page-locked host memory ----------------------------------------------------------------
// true if [p, p+n) lies inside one allocation of modgpu_host_alloc that is really page-locked.
bool host_range_pinned(const void *p, uint64_t n);
// a host-fed launch has happened on the calling thread (host_stream.cpp): path stats and modgpu_last_launch
void note_feed_launch(uint32_t grid, uint64_t bytes);

// ---- host-buffer / file endpoints (host_stream.cpp) ---------------------------------------
// Where a stream's bytes come from / go to: caller memory, or a file read / written at offsets
// (pread / pwrite: safe from several pipeline threads at once).
struct Endpoint {
    uint8_t *mem = nullptr; // if set, bytes live at mem[0..n)
    int fd = -1;            // else file descriptor, bytes at file offset base + [0..n)
    uint64_t base = 0;
    bool pinned = false;    // mem is page-locked and device-visible: DMA'd directly, no staging copy
};
struct Piece { uint64_t off, len; }; // a span of a stream, in stream bytes
// What a call did besides succeeding or failing.
struct StreamOutcome {
    bool touched = false;          // dst may differ from what it was: a caller with a second engine can no longer simply start over
    bool finished_on_host = false; // the GPU was lost after the call had begun and the host loop did the pieces whose result had
                                   // not reached dst (rc is MODGPU_OK then)
    uint64_t host_bytes = 0;       // ... that many bytes
};
// src -> GPU -> dst for n bytes.  host_may_finish: on a HIP failure after the call has begun, finish the undone pieces with the
// library's host loop where their plaintext still exists (host_stream.cpp: stream_impl's end says where it does not).
int stream_impl(const Endpoint &src, const Endpoint &dst, uint64_t n, int32_t key, uint64_t stream_off, int device,
                bool host_may_finish, StreamOutcome *out);
#ifdef MODGPU_TESTING_HOOKS
extern std::atomic<int> g_pinned_mode; // modgpu_debug_set_pinned_mode
extern std::atomic<int> g_staged_mode; // modgpu_debug_set_staged_mode
#endif

} // namespace modgpu
