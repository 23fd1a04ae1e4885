// tests/cpu_runtime_standin/hip/hip_runtime.h -- NOT part of the product, and NOT a compatibility layer for it.
//
// A CPU stand-in for the handful of HIP runtime calls libmodgpu.so's HOST code makes (modgpu_capi.cpp,
// host_stream.cpp), used by `make sanitize-lib` only: GPU AddressSanitizer / ThreadSanitizer do not exist on
// the pool, so the library's own threaded host code -- the staging pipelines' retire / refill state machine,
// the ticket ring, the host-range table, the per-thread error strings, the multi-device paths -- is built
// against this header with -fsanitize=address,undefined and -fsanitize=thread and driven by the CPU tests.
//
//   streams   real in-order queues, each drained by its own thread: work on two streams really overlaps,
//             so ThreadSanitizer sees the same interleavings the GPU runtime would produce
//   memory    device / pinned memory = ordinary host memory; copies = memcpy on the stream's thread
//   launch    the product's host loop on the span the launch plan describes (standin_launch.cpp) -- with the
//             work-queue shape's ticket-pair protocol emulated, so a pair handed to two overlapping launches is
//             detected (modgpu_shim_pair_collisions)
//   devices   MODGPU_SHIM_DEVICES of them (default 2): per-thread current device like HIP's
#pragma once
#include <cstddef>
#include <cstdint>

enum hipError_t : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorInvalidDevice = 101, hipErrorLaunchFailure = 719, hipErrorUnknown = 999 };
struct ShimStream;
struct ShimEvent;
using hipStream_t = ShimStream *;
using hipEvent_t = ShimEvent *;
enum hipStreamCaptureStatus : int { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
enum hipStreamCaptureMode : int { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
enum hipMemcpyKind : int { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipDeviceAttribute_t : int { hipDeviceAttributeMultiprocessorCount = 63, hipDeviceAttributeClockRate = 5 };
constexpr unsigned hipStreamNonBlocking = 1, hipHostMallocPortable = 1, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000, hipHostMallocDefault = 0,
                   hipHostRegisterPortable = 1, hipHostRegisterMapped = 2;

const char *hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int d);
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int d);
hipError_t hipDeviceSynchronize();
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned flags);
hipError_t hipHostRegister(void *p, size_t n, unsigned flags);
hipError_t hipHostUnregister(void *p);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *st);
hipError_t hipThreadExchangeStreamCaptureMode(hipStreamCaptureMode *m);
hipError_t hipMemset(void *p, int v, size_t n);
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipEventCreate(hipEvent_t *e);
constexpr unsigned hipEventDisableTiming = 2;
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags); // work queued on s after this waits for e's last record
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);

// shim internals shared with standin_launch.cpp
namespace shim {
void enqueue(hipStream_t s, void (*fn)(void *), void *arg); // runs fn(arg) on the stream's thread, in order; s == nullptr: the current device's null stream
}
extern "C" unsigned long long modgpu_shim_pair_collisions(void); // work-queue launches that found their ticket pair in use
extern "C" unsigned long long modgpu_shim_launches(int variant);  // launches the shim executed, by shape (0 small, 1 large, 2 queue)
