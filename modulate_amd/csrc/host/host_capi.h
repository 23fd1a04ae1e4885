// host_capi.h -- C entry points over the C++ host mirror (CEncryptionCycler, CArk) so that the
// Python test-suite and bench scripts can drive it with ctypes.  These are test/bench hooks of
// libmodulate_host.so; the drop-in boundary itself is include/modgpu.h.
// All functions return an eError ordinal (0 = eError_NoError) unless noted; -1 = C++ exception
// (text in modhost_last_error()).
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *modhost_last_error(void);
void modhost_select_platform(int ps4);                                                    /* CSettings::mbPS4 / msPlatform */
void modhost_set_flags(int overwrite_outputs, int ignore_new_files, int pack_all, int verbose);
void modhost_set_fix_quirks(int on);                                                      /* CSettings::mbFixReferenceQuirks */

/* CEncryptionCycler().Cycle(buf, n, key) on `device` (-1 = current). 0 ok, -1 threw. */
int modhost_cycle_via_class(uint8_t *buf, uint32_t n, int32_t key, int device);

/* Decode command (Modulate.cpp:452-502): <dir>/main_<platform>.hdr -> <same>.dec */
int modhost_decode(const char *dir);

/* CDtaFile: parse `in`, re-serialise into out (min(cap,size) bytes), text dump into dump (NUL-terminated, truncated to dump_cap) */
int modhost_dta_roundtrip(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t cap, uint64_t *size, char *dump, uint64_t dump_cap);

void *modhost_ark_new(void);
void modhost_ark_free(void *ark);
int modhost_ark_load(void *ark, const char *header_path);
int modhost_ark_parse_header(void *ark, const uint8_t *image, uint64_t n);
int modhost_ark_load_data(void *ark);
int modhost_ark_extract(void *ark, int first, int count, const char *target_dir);
int modhost_ark_construct_from_directory(void *ark, const char *input_dir, const void *reference_ark);
/* names: n NUL-terminated strings back to back */
int modhost_ark_construct_from_table(void *ark, const char *names, const uint32_t *sizes, int n, int n_arks, const char *ark_prefix);
int modhost_ark_build(void *ark, const char *input_dir);
int modhost_ark_build_from_memory(void *ark, const uint8_t *data, uint64_t n);
int modhost_ark_save(const void *ark, const char *output_dir, const char *header_name);
int modhost_ark_cycle_parts(void *ark, int32_t key, int n_devices);
void modhost_ark_enable_part_cipher(void *ark, int enable, int n_devices);
/* writes min(cap, size) bytes into out, full size into *size */
int modhost_ark_serialise_header(const void *ark, int encrypt, uint8_t *out, uint64_t cap, uint64_t *size);

int modhost_ark_num_files(const void *ark);
int modhost_ark_num_arks(const void *ark);
uint32_t modhost_ark_ark_size(const void *ark, int i);
const char *modhost_ark_ark_path(const void *ark, int i);
const char *modhost_ark_file_name(const void *ark, int i);
uint32_t modhost_ark_file_size(const void *ark, int i);
int64_t modhost_ark_file_offset(const void *ark, int i);
int modhost_ark_file_flags1(const void *ark, int i);
int modhost_ark_file_flags2(const void *ark, int i);
uint64_t modhost_ark_data_size(const void *ark);
const uint8_t *modhost_ark_data(const void *ark);
int modhost_ark_data_pinned(const void *ark);                                             /* 1: the part buffer is page-locked */

#ifdef __cplusplus
}
#endif
