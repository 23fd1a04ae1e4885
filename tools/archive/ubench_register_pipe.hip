// tools/ubench_register_pipe.hip -- host-buffer path variant: pin the caller's pages IN PLACE chunk by
// chunk (hipHostRegister) and DMA straight from / to them, instead of copying through pinned staging.
// T threads, each: register chunk -> H2D -> (kernel stand-in: none) -> D2H -> sync -> unregister.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (4ull << 30);
    char *host = (char *)aligned_alloc(4096, n);
    memset(host, 3, n);
    for (int threads : {1, 2, 4, 6, 8}) {
        for (size_t chunk : {(size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20}) {
            std::vector<char *> dev(threads * 2);
            std::vector<hipStream_t> st(threads * 2);
            for (int i = 0; i < threads * 2; ++i) { CHECK(hipMalloc((void **)&dev[i], chunk)); CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); }
            double best = 1e9;
            for (int rep = 0; rep < 2; ++rep) {
                double t0 = now();
                std::vector<std::thread> th;
                size_t n_chunks = (n + chunk - 1) / chunk;
                for (int t = 0; t < threads; ++t)
                    th.emplace_back([&, t] {
                        CHECK(hipSetDevice(0));
                        size_t prev[2] = {(size_t)-1, (size_t)-1};
                        size_t k = 0;
                        for (size_t c = t; c < n_chunks + 2 * threads; c += threads, ++k) {
                            int slot = (int)(k & 1);
                            if (prev[slot] != (size_t)-1) {
                                CHECK(hipStreamSynchronize(st[t * 2 + slot]));
                                CHECK(hipHostUnregister(host + prev[slot] * chunk));
                                prev[slot] = (size_t)-1;
                            }
                            if (c < n_chunks) {
                                size_t off = c * chunk, len = std::min(chunk, n - off);
                                CHECK(hipHostRegister(host + off, len, hipHostRegisterDefault));
                                CHECK(hipMemcpyAsync(dev[t * 2 + slot], host + off, len, hipMemcpyHostToDevice, st[t * 2 + slot]));
                                CHECK(hipMemcpyAsync(host + off, dev[t * 2 + slot], len, hipMemcpyDeviceToHost, st[t * 2 + slot]));
                                prev[slot] = c;
                            }
                        }
                    });
                for (auto &x : th) x.join();
                best = std::min(best, now() - t0);
            }
            printf("threads=%d chunk=%3zu MiB: %.4f s  %.1f GB/s payload\n", threads, chunk >> 20, best, n / best / 1e9);
            for (int i = 0; i < threads * 2; ++i) { CHECK(hipFree(dev[i])); CHECK(hipStreamDestroy(st[i])); }
        }
    }
    return 0;
}
