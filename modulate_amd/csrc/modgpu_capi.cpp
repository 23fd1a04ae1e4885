// modgpu_capi.cpp -- host side of the C ABI in include/modgpu.h.
//
// Compiled by hipcc as host-only C++ and linked with cycle_kernel.hip into libmodgpu.so.
// No CPU implementation of the cipher lives here: the only arithmetic done on the host is the
// per-launch jump-ahead (a handful of modular powers) that seeds the kernel.
#include "../../include/modgpu.h"

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cycle_kernel.h"
#include "lcg.h"

namespace {

thread_local std::string t_err;

int fail(int code, const char *what)
{
    t_err = what;
    return code;
}

int fail_hip(hipError_t e, const char *where)
{
    t_err = std::string(where) + ": " + hipGetErrorString(e);
    return MODGPU_ERR_HIP;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail_hip(e_, #expr);                                          \
    } while (0)

int device_count_raw()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// Makes `device` current for the calling thread (HIP's current device is per thread).
int select_device(int device)
{
    int n = device_count_raw();
    if (n <= 0) return fail(MODGPU_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0) return MODGPU_OK; // keep the thread's current device
    if (device >= n) return fail(MODGPU_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    return MODGPU_OK;
}

int resolve_device(int device, int *out)
{
    int rc = select_device(device);
    if (rc) return rc;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    *out = device;
    return MODGPU_OK;
}

// ---- launch planning -----------------------------------------------------------------

struct Plan {
    CycleArgs args;
    int variant;
    uint32_t grid;
};

// Streaming shape: one persistent 1024-thread workgroup per CU (4 waves/SIMD), so the grid is the
// device's CU count (256 on MI355X), looked up once per device.
uint32_t large_grid()
{
    static std::mutex mu;
    static uint32_t cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256u;
    std::lock_guard<std::mutex> lock(mu);
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = (uint32_t)std::min(n, 2048); // grid * 32 tiles must stay <= 65536
    }
    return cus[dev];
}
constexpr uint32_t kSmallGridMax = 16384u;   // 4 KiB chunks: grid * 1 tile <= 65536
// Hand-over between the two shapes, measured warm and cold (profiles/r01_tune_cycle_sizes*.txt):
// up to 256 MiB the one-shot 4 KiB-chunk grid wins (launch cost ~3 us vs ~9 us, and the buffer fits
// the 256 MiB Infinity Cache); beyond it the 128 KiB-burst streaming kernel does.
constexpr uint64_t kLargeMin = (256ull << 20) + 1;

// Splits [buf, buf+n) into <16 head bytes, an aligned body of 16-byte words and <16 tail bytes,
// and computes the states that seed each piece.  key_res != 0.
Plan plan_cycle(void *dev_buf, uint64_t n, uint32_t key_res, uint64_t stream_off)
{
    Plan p{};
    uintptr_t addr = reinterpret_cast<uintptr_t>(dev_buf);
    uint64_t head = std::min<uint64_t>(n, (16 - (addr & 15)) & 15);
    uint64_t words = (n - head) / 16;
    uint64_t tail = n - head - words * 16;
    // the stream position of byte j is stream_off + j; positions reduce mod PERIOD
    uint64_t o = stream_off % lcg::PERIOD;

    CycleArgs &a = p.args;
    a.head_ptr = static_cast<uint8_t *>(dev_buf);
    a.head_n = (uint32_t)head;
    a.body = a.head_ptr + head;
    a.body_words = words;
    a.tail_ptr = a.head_ptr + head + words * 16;
    a.tail_n = (uint32_t)tail;
    a.base_head = lcg::state_residue(key_res, o);
    a.base_body = lcg::state_residue(key_res, o + head);
    a.base_tail = lcg::state_residue(key_res, o + head + (words * 16) % lcg::PERIOD);

    uint64_t body_bytes = words * 16;
    p.variant = body_bytes >= kLargeMin ? CYCLE_LARGE : CYCLE_SMALL;
    // test / tuning knobs (read per call): MODGPU_FORCE_SHAPE=small|large picks the launch shape
    // whatever the size, MODGPU_GRID caps the grid -- together they let the test-suite drive the
    // streaming kernel through many trips and ragged ends on buffers of a few MiB.
    if (const char *f = std::getenv("MODGPU_FORCE_SHAPE")) {
        if (!std::strcmp(f, "large")) p.variant = CYCLE_LARGE;
        else if (!std::strcmp(f, "small")) p.variant = CYCLE_SMALL;
    }
    uint64_t chunk = modgpu_variant_chunk_bytes(p.variant);
    // chunks sit on absolute chunk-aligned addresses: the first starts `lead` bytes before the body,
    // and the kernel counts positions from there, so its base state is stepped back by a^(-lead)
    a.lead = (uint32_t)(reinterpret_cast<uintptr_t>(a.body) & (chunk - 1));
    a.base_body = lcg::mulmod(a.base_body, lcg::powmod(lcg::A, lcg::PERIOD - a.lead % lcg::PERIOD));
    uint64_t chunks = (a.lead + body_bytes + chunk - 1) / chunk;
    uint64_t cap = p.variant == CYCLE_LARGE ? large_grid() : kSmallGridMax;
    if (const char *g = std::getenv("MODGPU_GRID")) {
        long v = std::atol(g);
        if (v >= 1 && (uint64_t)v < cap) cap = (uint64_t)v;
    }
    p.grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(chunks, cap));
    // one grid trip advances every lane-word by grid chunks
    a.stride_mul2 = 2u * lcg::powmod(lcg::A, ((uint64_t)p.grid * chunk) % lcg::PERIOD);
    return p;
}

int cycle_device_impl(void *dev_buf, uint64_t n, int32_t key, uint64_t stream_off, hipStream_t stream)
{
    if (n == 0) return MODGPU_OK;
    if (!dev_buf) return fail(MODGPU_ERR_INVALID, "null device buffer");
    uint32_t key_res = lcg::key_residue(key);
    if (key_res == 0) return MODGPU_OK; // keystream is all zero (state sticks at m): identity
    Plan p = plan_cycle(dev_buf, n, key_res, stream_off);
    hipError_t e = modgpu_launch_cycle(p.args, p.variant, p.grid, stream);
    if (e != hipSuccess) return fail_hip(e, "cycle kernel launch");
    return MODGPU_OK;
}

// ---- per-device staging context for host-buffer calls ----------------------------------
//
// A caller-owned pageable buffer goes  memcpy -> pinned -> H2D -> kernel -> D2H -> pinned -> memcpy.
// Measured on the MI355X node (profiles/r01_ubench_hostpath.txt): the DMA engines move 57 GB/s each
// way from pinned memory, one host thread copies pageable->pinned at 22 GB/s, four at 73 GB/s, and
// registering the caller's pages in place costs as much as copying them.  So the work is spread
// over kPipes (4) independent pipelines, each a host thread with two (pinned, device, stream) slots
// that double-buffers its own chunks; the copies of different pipelines overlap each other and
// the (comparatively instant) kernels.  Small buffers use pipeline 0 inline, no threads.

constexpr int kMaxPipes = 16;
constexpr int kSlotsPerPipe = 2;
constexpr int kSlots = kMaxPipes * kSlotsPerPipe;

// Tunables (read once): MODGPU_HOST_PIPES = host threads / independent pipelines for large buffers,
// MODGPU_HOST_CHUNK_MB = bytes per slot in MiB.
int env_int(const char *name, int dflt, int lo, int hi)
{
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    int x = std::atoi(v);
    return x < lo ? lo : (x > hi ? hi : x);
}
const int kPipes = env_int("MODGPU_HOST_PIPES", 4, 1, kMaxPipes);
const uint64_t kChunk = (uint64_t)env_int("MODGPU_HOST_CHUNK_MB", 16, 1, 256) << 20;
// MODGPU_HOST_ZEROCOPY_KB: largest host buffer cycled in place in pinned memory by the kernel (0 = never)
const uint64_t kZeroCopyMax = (uint64_t)env_int("MODGPU_HOST_ZEROCOPY_KB", 1024, 0, 1 << 20) << 10;

struct Staging {
    std::mutex mu;
    uint8_t *pinned[kSlots] = {};
    uint8_t *dev[kSlots] = {};
    hipStream_t stream[kSlots] = {};
    uint64_t cap[kSlots] = {};
};

constexpr int kMaxDevices = 64;
Staging g_staging[kMaxDevices];

// Slots [0, n_slots) get at least `need` bytes each (grown on demand, never shrunk).
int staging_reserve(Staging &s, int n_slots, uint64_t need)
{
    need = std::min<uint64_t>(std::max<uint64_t>(need, 1ull << 20), kChunk);
    for (int i = 0; i < n_slots; ++i) {
        if (s.cap[i] >= need) continue;
        if (s.pinned[i]) HIP_TRY(hipHostFree(s.pinned[i]));
        if (s.dev[i]) HIP_TRY(hipFree(s.dev[i]));
        s.pinned[i] = nullptr;
        s.dev[i] = nullptr;
        s.cap[i] = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.pinned[i]), need, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s.dev[i]), need));
        if (!s.stream[i]) HIP_TRY(hipStreamCreateWithFlags(&s.stream[i], hipStreamNonBlocking));
        s.cap[i] = need;
    }
    return MODGPU_OK;
}

// Where a stream's bytes come from / go to: caller memory, or a file read / written at offsets
// (pread / pwrite: safe from several pipeline threads at once).
struct Endpoint {
    uint8_t *mem = nullptr; // if set, bytes live at mem[0..n)
    int fd = -1;            // else file descriptor, bytes at file offset base + [0..n)
    uint64_t base = 0;
};

int io_fail(const char *what)
{
    t_err = std::string(what) + ": " + std::strerror(errno);
    return MODGPU_ERR_IO;
}

int fill_slot(const Endpoint &src, uint8_t *pinned, uint64_t off, uint64_t len)
{
    if (src.mem) {
        std::memcpy(pinned, src.mem + off, len);
        return MODGPU_OK;
    }
    for (uint64_t done = 0; done < len;) {
        ssize_t r = ::pread(src.fd, pinned + done, len - done, (off_t)(src.base + off + done));
        if (r < 0 && errno == EINTR) continue;
        if (r < 0) return io_fail("pread");
        if (r == 0) return fail(MODGPU_ERR_IO, "pread: unexpected end of file");
        done += (uint64_t)r;
    }
    return MODGPU_OK;
}

int drain_slot(const Endpoint &dst, const uint8_t *pinned, uint64_t off, uint64_t len)
{
    if (dst.mem) {
        std::memcpy(dst.mem + off, pinned, len);
        return MODGPU_OK;
    }
    for (uint64_t done = 0; done < len;) {
        ssize_t r = ::pwrite(dst.fd, pinned + done, len - done, (off_t)(dst.base + off + done));
        if (r < 0 && errno == EINTR) continue;
        if (r < 0) return io_fail("pwrite");
        done += (uint64_t)r;
    }
    return MODGPU_OK;
}

// One pipeline: chunks first, first+stride, ... of the stream through slots [slot0, slot0+2).
int run_pipe(Staging &s, int slot0, const Endpoint &src, const Endpoint &dst, uint64_t n, uint64_t chunk,
             uint64_t first, uint64_t stride, int32_t key, uint64_t stream_off)
{
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    auto span = [&](uint64_t c, uint64_t *off, uint64_t *len) {
        *off = c * chunk;
        *len = std::min<uint64_t>(chunk, n - *off);
    };
    uint64_t mine = first < n_chunks ? (n_chunks - first + stride - 1) / stride : 0;
    int rc = MODGPU_OK;
    for (uint64_t i = 0; i < mine + kSlotsPerPipe; ++i) {
        int slot = slot0 + (int)(i % kSlotsPerPipe);
        if (i >= kSlotsPerPipe) { // drain the chunk that used this slot two trips ago
            uint64_t off, len;
            span(first + (i - kSlotsPerPipe) * stride, &off, &len);
            HIP_TRY(hipStreamSynchronize(s.stream[slot]));
            if (rc == MODGPU_OK) rc = drain_slot(dst, s.pinned[slot], off, len);
        }
        if (i < mine && rc == MODGPU_OK) {
            uint64_t off, len;
            span(first + i * stride, &off, &len);
            rc = fill_slot(src, s.pinned[slot], off, len);
            if (rc) continue; // keep draining what is already in flight, then report
            HIP_TRY(hipMemcpyAsync(s.dev[slot], s.pinned[slot], len, hipMemcpyHostToDevice, s.stream[slot]));
            rc = cycle_device_impl(s.dev[slot], len, key, stream_off + off, s.stream[slot]);
            if (rc) continue;
            HIP_TRY(hipMemcpyAsync(s.pinned[slot], s.dev[slot], len, hipMemcpyDeviceToHost, s.stream[slot]));
        }
    }
    return rc;
}

// src -> pinned -> H2D -> kernel -> D2H -> pinned -> dst for n bytes, over 1..kPipes pipelines.
int stream_impl(const Endpoint &src, const Endpoint &dst, uint64_t n, int32_t key, uint64_t stream_off, int device)
{
    if (n == 0) return MODGPU_OK;
    int dev = 0;
    int rc = resolve_device(device, &dev);
    if (rc) return rc;
    // keys == 0 mod m give the identity: nothing to do in place, a plain copy otherwise
    const bool identity = lcg::key_residue(key) == 0;
    if (identity && src.mem && src.mem == dst.mem) return MODGPU_OK;
    if (dev >= kMaxDevices) return fail(MODGPU_ERR_INVALID, "device index beyond staging table");
    Staging &s = g_staging[dev];
    std::lock_guard<std::mutex> lock(s.mu);

    // Header-sized buffers (what the reference's three call sites actually pass: <= 512 KiB): skip the two
    // DMA submissions and let the kernel read and write the pinned staging buffer across PCIe itself
    // (hipHostMalloc memory is device-visible).  One launch + one sync instead of copy + launch + copy.
    if (n <= kZeroCopyMax && src.mem && dst.mem && !identity) {
        rc = staging_reserve(s, 1, n);
        if (rc) return rc;
        void *mapped = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&mapped, s.pinned[0], 0));
        std::memcpy(s.pinned[0], src.mem, n);
        rc = cycle_device_impl(mapped, n, key, stream_off, s.stream[0]);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(s.stream[0]));
        std::memcpy(dst.mem, s.pinned[0], n);
        return MODGPU_OK;
    }

    // slot size: the whole buffer if it is small, else ~n/16 between 4 MiB and the cap (measured:
    // 4 MiB slots are best at 64 MiB, 16 MiB slots from 1 GiB up; profiles/r01_sweep_hostpath.txt)
    uint64_t chunk = n <= (4ull << 20) ? std::max<uint64_t>(n, 1ull << 20)
                                       : std::min<uint64_t>(kChunk, std::max<uint64_t>(4ull << 20, ((n >> 4) + 0xFFFFF) & ~0xFFFFFull));
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    const int pipes = (int)std::min<uint64_t>((uint64_t)kPipes, (n_chunks + 1) / 2); // a pipeline is worth >= 2 chunks
    rc = staging_reserve(s, pipes * kSlotsPerPipe, chunk);
    if (rc) return rc;
    if (pipes <= 1) return run_pipe(s, 0, src, dst, n, chunk, 0, 1, key, stream_off);

    std::vector<int> rcs(pipes, MODGPU_OK);
    std::vector<std::string> errs(pipes);
    std::vector<std::thread> workers;
    auto body = [&](int p) {
        if (hipSetDevice(dev) != hipSuccess) { // HIP's current device is per thread
            rcs[p] = MODGPU_ERR_HIP;
            errs[p] = "hipSetDevice in staging worker";
            return;
        }
        rcs[p] = run_pipe(s, p * kSlotsPerPipe, src, dst, n, chunk, (uint64_t)p, (uint64_t)pipes, key, stream_off);
        if (rcs[p]) errs[p] = t_err;
    };
    for (int p = 1; p < pipes; ++p) workers.emplace_back(body, p);
    body(0);
    for (auto &w : workers) w.join();
    for (int p = 0; p < pipes; ++p)
        if (rcs[p]) {
            t_err = errs[p];
            return rcs[p];
        }
    return MODGPU_OK;
}

int cycle_host_impl(uint8_t *host, uint64_t n, int32_t key, uint64_t stream_off, int device)
{
    if (n == 0) return MODGPU_OK;
    if (!host) return fail(MODGPU_ERR_INVALID, "null host buffer");
    Endpoint e;
    e.mem = host;
    return stream_impl(e, e, n, key, stream_off, device);
}

struct Fd { // closes on scope exit
    int fd = -1;
    ~Fd() { if (fd >= 0) ::close(fd); }
};

uint32_t load_le32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

void store_le32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

} // namespace

extern "C" {

int modgpu_abi_version(void) { return MODGPU_ABI_VERSION; }

int modgpu_device_count(void) { return device_count_raw(); }

const char *modgpu_last_error(void) { return t_err.c_str(); }

int modgpu_cycle_device(void *dev_buf, uint64_t n, int32_t key, uint64_t stream_off, int device,
                        void *hip_stream)
{
    int rc = select_device(device);
    if (rc) return rc;
    return cycle_device_impl(dev_buf, n, key, stream_off, static_cast<hipStream_t>(hip_stream));
}

int modgpu_cycle_host(uint8_t *host_buf, uint64_t n, int32_t key, uint64_t stream_off, int device)
{
    return cycle_host_impl(host_buf, n, key, stream_off, device);
}

int modgpu_hdr_decrypt_host(uint8_t *hdr, uint64_t size, int device)
{
    if (!hdr || size < 4) return fail(MODGPU_ERR_INVALID, "header shorter than its magic");
    uint32_t magic = load_le32(hdr);
    if (magic != MODGPU_MAGIC_PS3 && magic != MODGPU_MAGIC_PS4)
        return fail(MODGPU_ERR_MAGIC, "unknown header magic");
    uint32_t key = magic == MODGPU_MAGIC_PS3 ? MODGPU_KEY_PS3 : MODGPU_KEY_PS4;
    return cycle_host_impl(hdr + 4, size - 4, (int32_t)key, 0, device);
}

int modgpu_hdr_encrypt_host(uint8_t *hdr, uint64_t size, int ps4, int device)
{
    if (!hdr || size < 4) return fail(MODGPU_ERR_INVALID, "header shorter than its magic");
    // cipher first: on failure the caller's buffer is left as it was
    int rc = cycle_host_impl(hdr + 4, size - 4, (int32_t)(ps4 ? MODGPU_KEY_PS4 : MODGPU_KEY_PS3), 0, device);
    if (rc) return rc;
    store_le32(hdr, ps4 ? MODGPU_MAGIC_PS4 : MODGPU_MAGIC_PS3);
    return MODGPU_OK;
}

int modgpu_cycle_parts_host(uint8_t *const *parts, const uint64_t *sizes, int n_parts, int32_t key,
                            int n_devices)
{
    if (n_parts < 0 || (n_parts > 0 && (!parts || !sizes))) return fail(MODGPU_ERR_INVALID, "bad part list");
    int avail = device_count_raw();
    if (avail <= 0) return fail(MODGPU_ERR_NO_DEVICE, "no HIP device visible");
    if (n_devices <= 0 || n_devices > avail) n_devices = avail;
    n_devices = std::min(n_devices, std::max(n_parts, 1));
    std::vector<int> rcs(n_devices, MODGPU_OK);
    std::vector<std::string> errs(n_devices);
    std::vector<std::thread> workers;
    for (int d = 0; d < n_devices; ++d) {
        workers.emplace_back([&, d] {
            for (int i = d; i < n_parts; i += n_devices) { // part i -> GPU i mod N
                int rc = cycle_host_impl(parts[i], sizes[i], key, 0, d);
                if (rc) {
                    rcs[d] = rc;
                    errs[d] = t_err;
                    return;
                }
            }
        });
    }
    for (auto &w : workers) w.join();
    for (int d = 0; d < n_devices; ++d)
        if (rcs[d]) {
            t_err = errs[d];
            return rcs[d];
        }
    return MODGPU_OK;
}

int modgpu_cycle_file(const char *src_path, const char *dst_path, int32_t key, uint64_t stream_off, int device)
{
    if (!src_path || !dst_path) return fail(MODGPU_ERR_INVALID, "null path");
    const bool in_place = std::strcmp(src_path, dst_path) == 0;
    Fd in, out;
    in.fd = ::open(src_path, in_place ? O_RDWR : O_RDONLY);
    if (in.fd < 0) return io_fail(src_path);
    struct stat st;
    if (::fstat(in.fd, &st) != 0) return io_fail("fstat");
    if (!in_place) {
        out.fd = ::open(dst_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (out.fd < 0) return io_fail(dst_path);
    }
    Endpoint src, dst;
    src.fd = in.fd;
    dst.fd = in_place ? in.fd : out.fd;
    return stream_impl(src, dst, (uint64_t)st.st_size, key, stream_off, device);
}

int modgpu_cycle_file_to_host(const char *path, uint64_t file_off, uint8_t *host_dst, uint64_t n, int32_t key,
                              uint64_t stream_off, int device)
{
    if (!path || (n && !host_dst)) return fail(MODGPU_ERR_INVALID, "null path or buffer");
    Fd in;
    in.fd = ::open(path, O_RDONLY);
    if (in.fd < 0) return io_fail(path);
    Endpoint src, dst;
    src.fd = in.fd;
    src.base = file_off;
    dst.mem = host_dst;
    return stream_impl(src, dst, n, key, stream_off, device);
}

int modgpu_cycle_host_to_file(const uint8_t *host_src, uint64_t n, const char *path, int32_t key, uint64_t stream_off,
                              int device)
{
    if (!path || (n && !host_src)) return fail(MODGPU_ERR_INVALID, "null path or buffer");
    Fd out;
    out.fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (out.fd < 0) return io_fail(path);
    Endpoint src, dst;
    src.mem = const_cast<uint8_t *>(host_src); // only read from
    dst.fd = out.fd;
    return stream_impl(src, dst, n, key, stream_off, device);
}

int modgpu_alloc(void **dev_ptr, uint64_t n, int device)
{
    if (!dev_ptr) return fail(MODGPU_ERR_INVALID, "null out pointer");
    int rc = select_device(device);
    if (rc) return rc;
    HIP_TRY(hipMalloc(dev_ptr, n ? n : 1));
    return MODGPU_OK;
}

int modgpu_free(void *dev_ptr, int device)
{
    int rc = select_device(device);
    if (rc) return rc;
    HIP_TRY(hipFree(dev_ptr));
    return MODGPU_OK;
}

int modgpu_h2d(void *dev_dst, const void *host_src, uint64_t n, int device)
{
    int rc = select_device(device);
    if (rc) return rc;
    if (n) HIP_TRY(hipMemcpy(dev_dst, host_src, n, hipMemcpyHostToDevice));
    return MODGPU_OK;
}

int modgpu_d2h(void *host_dst, const void *dev_src, uint64_t n, int device)
{
    int rc = select_device(device);
    if (rc) return rc;
    if (n) HIP_TRY(hipMemcpy(host_dst, dev_src, n, hipMemcpyDeviceToHost));
    return MODGPU_OK;
}

int modgpu_sync(int device, void *hip_stream)
{
    int rc = select_device(device);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)));
    return MODGPU_OK;
}

int modgpu_time_cycle_device(void *dev_buf, uint64_t n, int32_t key, uint64_t stream_off, int device,
                             void *hip_stream, int iters, float *ms_per_launch)
{
    if (iters <= 0 || !ms_per_launch) return fail(MODGPU_ERR_INVALID, "bad timing arguments");
    int rc = select_device(device);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < iters && rc == MODGPU_OK; ++i) rc = cycle_device_impl(dev_buf, n, key, stream_off, st);
    hipError_t e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (e != hipSuccess) return fail_hip(e, "event timing");
    *ms_per_launch = ms / (float)iters;
    return MODGPU_OK;
}

uint32_t modgpu_state_at(int32_t key, uint64_t i)
{
    uint32_t r = lcg::state_residue(lcg::key_residue(key), i);
    return r ? r : lcg::M; // the reference shows residue 0 as m (CEncryptionCycler.cpp:19-22)
}

int modgpu_jump_table(int which, uint32_t *out, int count)
{
    if (!out || count < 0) return 0;
    const uint32_t *src = nullptr;
    int n = 0;
    switch (which) {
    case 0: src = lcg::kBytePow.v; n = 16; break;
    case 1: src = lcg::kLanePow.v; n = 256; break;
    case 2: src = lcg::kTileLo.v; n = 256; break;
    case 3: src = lcg::kTileHi.v; n = 256; break;
    default: return 0;
    }
    n = std::min(n, count);
    std::memcpy(out, src, (size_t)n * sizeof(uint32_t));
    return n;
}

} // extern "C"
