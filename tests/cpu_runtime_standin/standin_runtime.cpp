// standin_runtime.cpp -- see tests/cpu_runtime_standin/hip/hip_runtime.h.  CPU build for the sanitizers only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

struct ShimStream {
    std::mutex mu;
    std::condition_variable cv, idle;
    std::deque<std::pair<void (*)(void *), void *>> q;
    bool busy = false, quit = false;
    std::thread worker;
    ShimStream()
    {
        worker = std::thread([this] {
            std::unique_lock<std::mutex> lock(mu);
            for (;;) {
                cv.wait(lock, [this] { return quit || !q.empty(); });
                if (q.empty()) return;
                auto job = q.front();
                q.pop_front();
                busy = true;
                lock.unlock();
                job.first(job.second);
                lock.lock();
                busy = false;
                if (q.empty()) idle.notify_all();
            }
        });
    }
    void push(void (*fn)(void *), void *arg)
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            q.emplace_back(fn, arg);
        }
        cv.notify_one();
    }
    bool is_idle()
    {
        std::lock_guard<std::mutex> lock(mu);
        return q.empty() && !busy;
    }
    void drain()
    {
        std::unique_lock<std::mutex> lock(mu);
        idle.wait(lock, [this] { return q.empty() && !busy; });
    }
    ~ShimStream()
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            quit = true;
        }
        cv.notify_one();
        worker.join();
    }
};

struct ShimEvent {
    std::mutex mu;
    std::condition_variable cv;
    bool pending = false;
    std::chrono::steady_clock::time_point at{};
};

namespace {
constexpr int kMaxDev = 16;
int device_count()
{
    static const int n = [] {
        const char *e = std::getenv("MODGPU_SHIM_DEVICES");
        int v = e ? std::atoi(e) : 2;
        return v < 0 ? 0 : (v > kMaxDev ? kMaxDev : v);
    }();
    return n;
}
thread_local int t_device = 0;
std::mutex g_null_mu;
ShimStream *g_null[kMaxDev] = {};
ShimStream *null_stream()
{
    std::lock_guard<std::mutex> lock(g_null_mu);
    if (!g_null[t_device]) g_null[t_device] = new ShimStream; // lives as long as the process (like HIP's)
    return g_null[t_device];
}
ShimStream *resolve(hipStream_t s) { return s ? s : null_stream(); }

struct CopyJob { void *dst; const void *src; size_t n; int fill; bool is_set; };
void run_copy(void *p)
{
    CopyJob *j = static_cast<CopyJob *>(p);
    if (j->is_set) std::memset(j->dst, j->fill, j->n);
    else if (j->dst != j->src) std::memmove(j->dst, j->src, j->n);
    delete j;
}
void run_event(void *p)
{
    ShimEvent *e = static_cast<ShimEvent *>(p);
    std::lock_guard<std::mutex> lock(e->mu);
    e->at = std::chrono::steady_clock::now();
    e->pending = false;
    e->cv.notify_all();
}
} // namespace

namespace shim {
void enqueue(hipStream_t s, void (*fn)(void *), void *arg) { resolve(s)->push(fn, arg); }
} // namespace shim

const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorInvalidDevice ? "invalid device ordinal (shim)" : "error (shim)"); }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = device_count(); return hipSuccess; }
hipError_t hipSetDevice(int d)
{
    if (d < 0 || d >= device_count()) return hipErrorInvalidDevice;
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d)
{
    if (device_count() == 0) return hipErrorInvalidDevice;
    *d = t_device;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeClockRate ? 2400000 : 8; return hipSuccess; } // 8 "CUs": small grids, many trips
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int d) { std::snprintf(buf, (size_t)len, "0000:%02X:00.0", 0x10 + d); return hipSuccess; }
hipError_t hipDeviceSynchronize() { null_stream()->drain(); return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return posix_memalign(p, 4096, n ? n : 1) == 0 ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new ShimStream; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { resolve(s)->drain(); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { return resolve(s)->is_idle() ? hipSuccess : hipErrorNotReady; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipThreadExchangeStreamCaptureMode(hipStreamCaptureMode *) { return hipSuccess; }
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s) { resolve(s)->push(run_copy, new CopyJob{p, nullptr, n, v, true}); return hipSuccess; }
hipError_t hipMemset(void *p, int v, size_t n) { hipMemsetAsync(p, v, n, nullptr); return hipStreamSynchronize(nullptr); }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s) { resolve(s)->push(run_copy, new CopyJob{dst, src, n, 0, false}); return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k) { hipMemcpyAsync(dst, src, n, k, nullptr); return hipStreamSynchronize(nullptr); }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new ShimEvent; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    {
        std::lock_guard<std::mutex> lock(e->mu);
        e->pending = true;
    }
    resolve(s)->push(run_event, e);
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    std::unique_lock<std::mutex> lock(e->mu);
    e->cv.wait(lock, [e] { return !e->pending; });
    return hipSuccess;
}
static void run_wait_event(void *p) { (void)hipEventSynchronize(static_cast<hipEvent_t>(p)); }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    resolve(s)->push(run_wait_event, e);
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->at - a->at).count();
    return hipSuccess;
}

// what tests/san_lib_cases.py needs of the runtime itself, with C linkage for ctypes
extern "C" {
int modgpu_shim_stream_create(void **s) { return (int)hipStreamCreateWithFlags(reinterpret_cast<hipStream_t *>(s), hipStreamNonBlocking); }
int modgpu_shim_stream_destroy(void *s) { return (int)hipStreamDestroy(static_cast<hipStream_t>(s)); }
int modgpu_shim_set_device(int d) { return (int)hipSetDevice(d); }
int modgpu_shim_get_device(void)
{
    int d = -1;
    return hipGetDevice(&d) == hipSuccess ? d : -1;
}
}
