// scalar_path.cpp -- the library's own HOST implementation of the cipher, for machines without a
// usable GPU (the reference's Cycle "cannot fail", Modulate/CEncryptionCycler.cpp:4-14).
//
// Product code: built from lcg.h's closed form  s_i = a^(i+1) * key mod m  like the kernel, not
// from the reference's Schrage step, and it shares nothing with the checker under oracle/.  No HIP
// in this file, so the sanitizer build (make sanitize) compiles it as is.
//
// Layout mirrors one GPU lane-word: sixteen independent byte states S_j = S * a^j, each advanced by
// a^16 per 16-byte word, so there is no serial dependency between neighbouring bytes and the
// compiler can keep the sixteen 32x32->64 multiplies in vector registers.  Buffers of 4 MiB and up
// are cut into contiguous spans, one per host thread (every span jumps to its own position).
#include "scalar_path.h"

#include <algorithm>
#include <thread>
#include <vector>

#include "lcg.h"

namespace {

// x, y < 2^31  ->  x*y mod m, in [0, m).  2^31 == 1 (mod m): fold the 62-bit product once.
inline uint32_t mulmod_fold(uint32_t x, uint32_t y)
{
    uint64_t p = (uint64_t)x * y;
    uint32_t r = (uint32_t)(p & lcg::M) + (uint32_t)(p >> 31); // < 2m
    return r >= lcg::M ? r - lcg::M : r;
}

void span_cycle(uint8_t *buf, uint64_t n, uint32_t key_res, uint64_t pos)
{
    constexpr int W = lcg::WORD;
    const uint32_t step = lcg::mulmod(lcg::kBytePow.v[W - 1], lcg::A); // a^16
    uint32_t s[W];
    s[0] = lcg::state_residue(key_res, pos);
    for (int j = 1; j < W; ++j) s[j] = mulmod_fold(s[j - 1], lcg::A);
    uint64_t i = 0;
    for (; i + W <= n; i += W) {
        uint8_t ks[W];
        for (int j = 0; j < W; ++j) {
            ks[j] = (uint8_t)~s[j]; // states are canonical and never 0 here (key_res != 0, m prime)
            s[j] = mulmod_fold(s[j], step);
        }
        for (int j = 0; j < W; ++j) buf[i + j] ^= ks[j];
    }
    for (int j = 0; i < n; ++i, ++j) buf[i] ^= (uint8_t)~s[j];
}

} // namespace

void modgpu_scalar_cycle(uint8_t *buf, uint64_t n, int32_t key, uint64_t stream_off)
{
    const uint32_t key_res = lcg::key_residue(key);
    if (n == 0 || key_res == 0) return; // residue 0 sticks at m: keystream all zero (identity)
    const uint64_t pos = stream_off % lcg::PERIOD;
    constexpr uint64_t kSpanMin = 4ull << 20;
    unsigned hw = std::thread::hardware_concurrency();
    uint64_t threads = std::min<uint64_t>(std::max(1u, std::min(hw, 32u)), n / kSpanMin);
    if (threads <= 1) {
        span_cycle(buf, n, key_res, pos);
        return;
    }
    const uint64_t per = ((n + threads - 1) / threads + 63) & ~63ull;
    std::vector<std::thread> pool;
    try {
        for (uint64_t off = per; off < n; off += per)
            pool.emplace_back(span_cycle, buf + off, std::min(per, n - off), key_res, pos + off % lcg::PERIOD);
    } catch (...) { // thread limit reached: finish what was not handed out on this thread
        uint64_t done = per * (pool.size() + 1);
        if (done < n) span_cycle(buf + done, n - done, key_res, pos + done % lcg::PERIOD);
    }
    span_cycle(buf, std::min(per, n), key_res, pos);
    for (auto &t : pool) t.join();
}
