/*
 * modgpu.h -- C ABI of the MI355X (gfx950) implementation of Modulate's cipher hot path.
 *
 * The reference (AdamClixby/Modulate) has no FFI or plugin interface; the seam it offers is
 * one C++ class,
 *
 *     class CEncryptionCycler { public: void Cycle(unsigned char*, unsigned int, int); ... };
 *                                                  (Modulate/CEncryptionCycler.h:3-10)
 *
 * called from exactly three places, always as  Cycle(buf + 4, size - 4, key):
 *     Modulate/CArk.cpp:338-339     CArk::Load            (header decrypt)
 *     Modulate/CArk.cpp:1135-1136   CArk::SaveArk         (header encrypt)
 *     Modulate/Modulate.cpp:485-486 Decode                (-decode command)
 *
 * This header is what that class's body binds to (modulate_amd/csrc/host/CEncryptionCycler.cpp is
 * the binding; INTEGRATION.md shows the same stub for the upstream tree).  Plain pointers and
 * sizes only; no C++ or torch types.  Every function returns MODGPU_OK (0) or a MODGPU_ERR_*
 * code, never throws, never prints; modgpu_last_error() gives the text for the calling thread.
 *
 * Semantics (bit-exact with Modulate/CEncryptionCycler.cpp:4-25):
 *     ks[i]  = low8( a^(i+1) * key mod (2^31-1) ) ^ 0xFF        a = 16807, residue 0 -> 2^31-1
 *     buf[j] ^= ks[stream_off + j]                               j = 0 .. n-1
 * stream_off = 0 reproduces one reference Cycle call.  n and stream_off are 64-bit, which lifts
 * the reference's `unsigned int` length cap (2^32-1) and lets one logical stream be split over
 * calls or devices.  Keys congruent to 0 mod 2^31-1 give the identity, as in the reference.
 *
 * There is NO CPU implementation behind these entry points: without a usable HIP device the
 * compute calls fail with MODGPU_ERR_NO_DEVICE / MODGPU_ERR_HIP.
 *
 * Threading: callable concurrently from any number of host threads.  `device` selects the GPU
 * per call (-1 = the calling thread's current HIP device); no global "current device" is
 * relied on.  Host-buffer calls to the same device serialise on that device's staging context.
 */
#ifndef MODGPU_H
#define MODGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODGPU_OK 0
#define MODGPU_ERR_INVALID 1   /* bad argument (null pointer with n > 0, bad device index ...) */
#define MODGPU_ERR_NO_DEVICE 2 /* no HIP device visible                                        */
#define MODGPU_ERR_HIP 3       /* a HIP runtime call failed; see modgpu_last_error()           */
#define MODGPU_ERR_MAGIC 4     /* header magic is neither PS3 nor PS4 (eError_UnknownVersionNumber,
                                  Modulate/CArk.cpp:329-334, Modulate/Modulate.cpp:476-481)    */
#define MODGPU_ERR_IO 5        /* open / read / write of a part file failed (eError_FailedToOpenFile,
                                  eError_FailedToWriteData at Modulate/CArk.cpp:745-749, 883-889)     */

/* Settings.h:16-20 */
#define MODGPU_MAGIC_PS3 0xc64eed30u
#define MODGPU_MAGIC_PS4 0x6f303f55u
#define MODGPU_KEY_PS3 0xc64eed30u
#define MODGPU_KEY_PS4 0x90cfc0abu

/* ABI version of this header (bumped on any signature change). */
#define MODGPU_ABI_VERSION 2
int modgpu_abi_version(void);

/* Number of HIP devices visible to this process (0 if none / runtime unusable). */
int modgpu_device_count(void);

/* Text of the last error raised on the calling thread ("" if none).  Never NULL. */
const char *modgpu_last_error(void);

/* ---- the hot path ------------------------------------------------------------------ */

/* Replaces the loop body of CEncryptionCycler::Cycle (CEncryptionCycler.cpp:9-13) for a buffer
 * that is already device-resident.  `dev_buf` may have any byte alignment (the reference's
 * callers pass buf+4).  Asynchronous on `hip_stream` (a hipStream_t; NULL = the device's
 * null stream); the caller synchronises.  This is the entry point the roofline is measured on. */
int modgpu_cycle_device(void *dev_buf, uint64_t n, int32_t key, uint64_t stream_off,
                        int device, void *hip_stream);

/* Replaces CEncryptionCycler::Cycle (CEncryptionCycler.cpp:4-14) for a caller-owned HOST buffer:
 * H2D -> kernel -> D2H through pinned staging owned by this library, chunked and overlapped.
 * Synchronous: on return host_buf holds the result.  Never retains or frees host_buf. */
int modgpu_cycle_host(uint8_t *host_buf, uint64_t n, int32_t key, uint64_t stream_off, int device);

/* Header framing of CArk::Load (CArk.cpp:328-339) and Decode (Modulate.cpp:475-486):
 * LE u32 magic at hdr[0..3] selects the key, the cipher covers hdr[4..size).  Host buffer. */
int modgpu_hdr_decrypt_host(uint8_t *hdr, uint64_t size, int device);

/* Header framing of SaveArk (CArk.cpp:914-915, 1135-1136): stores the platform magic at
 * hdr[0..3] (ps4 != 0 -> PS4) and encrypts hdr[4..size) with the platform key.  Host buffer. */
int modgpu_hdr_encrypt_host(uint8_t *hdr, uint64_t size, int ps4, int device);

/* Part-level sharding beside CArk::LoadArkData / lSaveArk (CArk.cpp:723-758, 845-899): part i
 * is an independent stream (its own Cycle from offset 0) and goes to GPU  i mod n_devices,
 * one host thread per GPU, no inter-GPU traffic.  n_devices <= 0 means all visible devices. */
int modgpu_cycle_parts_host(uint8_t *const *parts, const uint64_t *sizes, int n_parts,
                            int32_t key, int n_devices);

/* ---- part files streamed through the GPU ----------------------------
 * The reference reads a part with one fread into the concatenated buffer (CArk.cpp:751) and writes
 * a slice with one fwrite (CArk.cpp:883).  These do the same transfers with the cipher applied on
 * the way, overlapped: pread -> pinned -> H2D -> kernel -> D2H -> pinned -> pwrite / caller memory,
 * several chunks in flight, without a pageable staging copy.  Each call is one stream whose first
 * byte has keystream position stream_off (0 = a part's own Cycle). */

/* Whole file src_path -> dst_path (created / truncated).  The two may be the same path (in place). */
int modgpu_cycle_file(const char *src_path, const char *dst_path, int32_t key, uint64_t stream_off, int device);

/* n bytes at byte offset file_off of `path` -> host_dst[0..n). */
int modgpu_cycle_file_to_host(const char *path, uint64_t file_off, uint8_t *host_dst, uint64_t n, int32_t key,
                              uint64_t stream_off, int device);

/* host_src[0..n) -> `path` (created / truncated).  host_src is not modified. */
int modgpu_cycle_host_to_file(const uint8_t *host_src, uint64_t n, const char *path, int32_t key, uint64_t stream_off,
                              int device);

/* ---- thin device-memory helpers (bench / tests / callers that keep parts resident) --- */
int modgpu_alloc(void **dev_ptr, uint64_t n, int device);
int modgpu_free(void *dev_ptr, int device);
int modgpu_h2d(void *dev_dst, const void *host_src, uint64_t n, int device);
int modgpu_d2h(void *host_dst, const void *dev_src, uint64_t n, int device);
int modgpu_sync(int device, void *hip_stream);

/* Runs `iters` back-to-back modgpu_cycle_device launches on `hip_stream` bracketed by HIP events
 * recorded on that same stream and returns the mean milliseconds per launch in *ms_per_launch
 * (an even `iters` leaves the buffer unchanged: the cipher is an involution). */
int modgpu_time_cycle_device(void *dev_buf, uint64_t n, int32_t key, uint64_t stream_off,
                             int device, void *hip_stream, int iters, float *ms_per_launch);

/* ---- host-side jump-ahead arithmetic (exposed so it can be checked without a GPU) ---- */

/* State the reference loop holds when it XORs stream byte i: a^(i+1)*key mod m, in [1, m];
 * this is what the library feeds the kernel as its per-launch base. */
uint32_t modgpu_state_at(int32_t key, uint64_t i);

/* Fills out[0..count) with the kernel's compile-time jump tables so tests can verify them
 * against independent arithmetic.  which: 0 = a^j (j<16), 1 = a^(16*t) (t<256),
 * 2 = a^(4096*b) (b<256), 3 = a^(4096*256*b) (b<256).  Returns entries written. */
int modgpu_jump_table(int which, uint32_t *out, int count);

#ifdef __cplusplus
}
#endif
#endif /* MODGPU_H */
