// numa_place.cpp -- see numa_place.h.  Linux sysfs + the mbind / sched_setaffinity system calls; no libnuma.
#include "numa_place.h"

#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <mutex>
#include <thread>

namespace modgpu {
namespace numa {

bool enabled()
{
    static const bool v = [] {
        const char *e = std::getenv("MODGPU_NUMA");
        return !(e && std::strcmp(e, "0") == 0);
    }();
    return v;
}

std::vector<int> parse_cpulist(const std::string &text)
{
    std::vector<int> out;
    size_t i = 0;
    auto number = [&](long *v) {
        size_t start = i;
        long x = 0;
        while (i < text.size() && text[i] >= '0' && text[i] <= '9') x = x * 10 + (text[i++] - '0');
        *v = x;
        return i > start;
    };
    while (i < text.size()) {
        long lo = 0, hi = 0;
        if (!number(&lo)) { // not a digit: separator, newline, garbage
            ++i;
            continue;
        }
        hi = lo;
        if (i < text.size() && text[i] == '-') {
            ++i;
            if (!number(&hi)) continue;
        }
        for (long c = lo; c <= hi && c - lo < 4096; ++c) out.push_back((int)c);
    }
    return out;
}

static bool read_text(const std::string &path, std::string *out)
{
    std::ifstream f(path);
    if (!f) return false;
    std::getline(f, *out, '\0');
    return true;
}

int node_of_pci(const std::string &sysfs, const std::string &bdf)
{
    std::string text;
    if (bdf.empty() || !read_text(sysfs + "/bus/pci/devices/" + bdf + "/numa_node", &text)) return -1;
    char *end = nullptr;
    long v = std::strtol(text.c_str(), &end, 10);
    return end == text.c_str() || v < 0 ? -1 : (int)v;
}

std::vector<int> cpus_of_node(const std::string &sysfs, int node)
{
    std::string text;
    if (node < 0 || !read_text(sysfs + "/devices/system/node/node" + std::to_string(node) + "/cpulist", &text)) return {};
    return parse_cpulist(text);
}

void *reserve(size_t bytes)
{
    void *p = ::mmap(nullptr, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    return p == MAP_FAILED ? nullptr : p;
}

void release(void *p, size_t bytes)
{
    if (p) ::munmap(p, bytes ? bytes : 1);
}

int prefer_node(void *p, size_t bytes, int node)
{
    if (node < 0 || node >= 1024 || !p || bytes == 0) return -1;
    const uintptr_t page = (uintptr_t)::sysconf(_SC_PAGESIZE);
    uintptr_t lo = ((uintptr_t)p + page - 1) & ~(page - 1), hi = ((uintptr_t)p + bytes) & ~(page - 1);
    if (hi <= lo) return -1;
    unsigned long mask[1024 / (8 * sizeof(unsigned long))] = {};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    constexpr int kMpolPreferred = 1;
    return (int)::syscall(SYS_mbind, (void *)lo, (unsigned long)(hi - lo), kMpolPreferred, mask, 1024ul + 1, 0u) == 0 ? 0 : -1;
}

void prefault(void *p, size_t bytes, int threads, const std::string &sysfs, int node)
{
    if (!p || bytes == 0) return;
    const size_t page = (size_t)::sysconf(_SC_PAGESIZE);
    const size_t per_thread_min = 64u << 20;
    size_t n = std::min<size_t>((size_t)std::max(threads, 1), (bytes + per_thread_min - 1) / per_thread_min);
    const size_t per = ((bytes + n - 1) / n + page - 1) & ~(page - 1);
    auto touch = [=](size_t lo, size_t hi, bool own_thread) {
        if (own_thread && node >= 0) (void)run_on_node(sysfs, node);
        volatile unsigned char *b = static_cast<volatile unsigned char *>(p);
        for (size_t o = lo; o < hi; o += page) b[o] = 0;
    };
    std::vector<std::thread> pool;
    try {
        for (size_t t = 1; t < n; ++t) pool.emplace_back(touch, t * per, std::min(bytes, (t + 1) * per), true);
    } catch (...) { // thread limit: the rest is touched below, by the lock call itself
    }
    touch(0, std::min(bytes, per), false);
    for (auto &t : pool) t.join();
}

int run_on_node(const std::string &sysfs, int node)
{
    const std::vector<int> cpus = cpus_of_node(sysfs, node);
    if (cpus.empty()) return -1;
    cpu_set_t now, want;
    CPU_ZERO(&want);
    if (::sched_getaffinity(0, sizeof now, &now) != 0) return -1;
    int n = 0;
    for (int c : cpus)
        if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &now)) { // never widen what the process was given
            CPU_SET(c, &want);
            ++n;
        }
    return n > 0 && ::sched_setaffinity(0, sizeof want, &want) == 0 ? 0 : -1;
}

int node_of_address(const void *p)
{
    if (!p) return -1;
    int node = -1;
    // (no libnuma in the image: the system call itself; 3 = MPOL_F_NODE | MPOL_F_ADDR)
    if (::syscall(SYS_get_mempolicy, &node, nullptr, 0ul, const_cast<void *>(p), 3ul) != 0) return -1;
    return node;
}

int node_of_file_page(int fd, unsigned long long offset)
{
    struct stat st;
    if (fd < 0 || ::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (unsigned long long)st.st_size <= offset) return -1;
    const unsigned long long page = (unsigned long long)::sysconf(_SC_PAGESIZE);
    void *m = ::mmap(nullptr, (size_t)page, PROT_READ, MAP_SHARED, fd, (off_t)(offset & ~(page - 1)));
    if (m == MAP_FAILED) return -1;
    const int node = node_of_address(m);
    ::munmap(m, (size_t)page);
    return node;
}

int move_to_node(int node, const void *allowed_cpu_set, size_t set_bytes)
{
    constexpr int kMaxNodes = 64;
    static std::mutex mu;
    static std::vector<int> cached[kMaxNodes];
    static bool known[kMaxNodes] = {};
    if (node < 0 || node >= kMaxNodes || !allowed_cpu_set || set_bytes < sizeof(cpu_set_t)) return -1;
    std::vector<int> cpus;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!known[node]) {
            cached[node] = cpus_of_node("/sys", node);
            known[node] = true;
        }
        cpus = cached[node];
    }
    const cpu_set_t *allowed = static_cast<const cpu_set_t *>(allowed_cpu_set);
    cpu_set_t want;
    CPU_ZERO(&want);
    int n = 0;
    for (int c : cpus)
        if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, allowed)) {
            CPU_SET(c, &want);
            ++n;
        }
    return n > 0 && ::sched_setaffinity(0, sizeof want, &want) == 0 ? 0 : -1;
}

} // namespace numa
} // namespace modgpu
