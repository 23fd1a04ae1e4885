// numa_place.h -- where host memory and host threads sit relative to a GPU (numa_place.cpp).  No HIP.
//
// The part path is the only multi-GPU path that shares a resource: host DRAM and the PCIe root
// complexes.  On a two-socket 8-GPU node a part's pages and the threads that copy them
// should live on the socket its GPU hangs off; everything here is best effort -- when the topology
// cannot be read, or MODGPU_NUMA=0, nothing is bound and everything still works.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

namespace modgpu {
namespace numa {

bool enabled(); // MODGPU_NUMA is not "0" (read once)

// "0-3,8,10-11" -> {0,1,2,3,8,10,11}.  Malformed pieces are skipped.
std::vector<int> parse_cpulist(const std::string &text);

// NUMA node of PCI function `bdf` ("0000:c1:00.0"): <sysfs>/bus/pci/devices/<bdf>/numa_node.  -1 = unknown
// (file missing, or the kernel's own -1 on single-node machines).
int node_of_pci(const std::string &sysfs, const std::string &bdf);

// CPUs of a node: <sysfs>/devices/system/node/node<N>/cpulist.  Empty = unknown.
std::vector<int> cpus_of_node(const std::string &sysfs, int node);

// Page-aligned anonymous memory whose pages will be faulted in on `node` (mbind, MPOL_PREFERRED: falls back to
// other nodes rather than failing).  node < 0: no policy.  nullptr on failure.  Release with release().
void *reserve(size_t bytes);
void release(void *p, size_t bytes);
// Binds the whole pages inside [p, p + bytes) to `node`; call before the pages are first touched.  0 = done.
int prefer_node(void *p, size_t bytes, int node);

// Touches every page of [p, p + bytes) from up to `threads` threads (each on `node`'s CPUs when node >= 0), so that a
// page-locking call that follows finds the pages present: first touch -- allocating and zeroing -- is what page-locking
// a fresh range spends its time on, and it parallelises; the locking itself does not.
void prefault(void *p, size_t bytes, int threads, const std::string &sysfs, int node);

// Restricts the calling thread to the CPUs of `node` (sched_setaffinity).  0 = done, -1 = left as it was.
int run_on_node(const std::string &sysfs, int node);

// NUMA node the page holding `p` lives on (get_mempolicy; a page not yet present is faulted in by the question).  -1 = unknown.
int node_of_address(const void *p);
// NUMA node of the page-cache page that holds byte `offset` of the open file `fd` (the page is mapped for a moment and asked about;
// a page not in the cache is read by the question, and then lives wherever the kernel put it).  -1 = unknown / beyond the file's end.
int node_of_file_page(int fd, unsigned long long offset);
// Moves the calling (worker) thread to the CPUs of `node` that `allowed` -- the affinity mask of the thread it works for --
// contains; unlike run_on_node this can take a thread from one node to another.  The node's CPU list is read from /sys once per
// node.  0 = done, -1 = left as it was.
int move_to_node(int node, const void *allowed_cpu_set, size_t set_bytes);

} // namespace numa
} // namespace modgpu
