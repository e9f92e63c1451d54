/*
 * stralg_compat.h -- the reference-named entry points of the suffix-array /
 * BWT-table construction path, exported by libstralg_amd.so with the exact
 * signatures and struct layouts of mailund/stralg, so that the library can
 * replace the reference's translation units for this path at link time
 * (the reference has no plugin registry; its boundary is symbol replacement
 * in libstralg, stralg/CMakeLists.txt:2-32 -- see INTEGRATION.md).
 *
 * Each declaration cites the reference declaration it replaces.
 * Ownership and error behaviour follow the reference: arrays handed back are
 * malloc/calloc memory that callers release with free(); sa->string is
 * borrowed; constructors never return NULL (on a device error they print the
 * reason to stderr and abort() -- there is no CPU fallback).
 */
#ifndef STRALG_COMPAT_H
#define STRALG_COMPAT_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* stralg/suffix_array.h:10-20 (40 bytes on LP64) */
struct suffix_array {
    uint8_t *string;   /* borrowed, NUL-terminated */
    uint32_t length;   /* strlen + 1 */
    uint32_t *array;   /* length entries, malloc'd */
    uint32_t *inverse; /* NULL after construction */
    uint32_t *lcp;     /* NULL after construction */
};

/* stralg/remap.h:9-19 (388 bytes) */
struct remap_table {
    uint32_t alphabet_size;
    signed char table[256];
    signed char rev_table[128];
};

/* stralg/bwt.h:36-44 (56 bytes) */
struct bwt_table {
    struct remap_table *remap_table;
    struct suffix_array *sa;
    uint32_t *c_table;
    uint32_t *o_table;     /* position-major: o_table[i * sigma + a] (bwt.c:50-57) */
    uint32_t **o_indices;  /* o_indices[i] = o_table + sigma * i */
    uint32_t *ro_table;
    uint32_t **ro_indices;
};

/* stralg/suffix_array_internal.h:10 */
struct suffix_array *allocate_sa_(uint8_t *string);

/* stralg/suffix_array.h:31-35 (sa_is.c:466-509) */
struct suffix_array *sa_is_construction(uint8_t *remapped_string, uint32_t alphabet_size);
/* stralg/suffix_array.h:37-41 (sa_is_mem.c:471-494): same array, same device path */
struct suffix_array *sa_is_mem_construction(uint8_t *remapped_string, uint32_t alphabet_size);
/* stralg/suffix_array.h:26-29 (skew.c:388-395): any bytes 1..255, not remapped */
struct suffix_array *skew_sa_construction(uint8_t *string);
/* stralg/suffix_array.h:43-51 (suffix_array.c:12-24) */
void free_suffix_array(struct suffix_array *sa);
void free_complete_suffix_array(struct suffix_array *sa);

/* stralg/suffix_array.h:94-99 (suffix_array.c:53-85): fill sa->inverse / sa->lcp (malloc'd; no-ops when
 * already present).  SURVEY.md section 8f "next" row 3: computed on the device. */
void compute_inverse(struct suffix_array *sa);
void compute_lcp(struct suffix_array *sa);

/* stralg/remap.h:21-33,43-47,80-86 (remap.c:8-114,155-165) */
struct remap_table *alloc_remap_table(const uint8_t *string);
void init_remap_table(struct remap_table *table, const uint8_t *string);
void dealloc_remap_table(struct remap_table *table);
void free_remap_table(struct remap_table *table);
uint8_t *remap(uint8_t *output, const uint8_t *input, struct remap_table *table);
uint8_t *remap_between(uint8_t *output, const uint8_t *from, const uint8_t *to, struct remap_table *table);
uint8_t *remap_between0(uint8_t *output, const uint8_t *from, const uint8_t *to, struct remap_table *table);
uint32_t remap_string(uint8_t *output, uint8_t *input);

/* stralg/bwt.h:73-76 (bwt.c:22-89) */
void init_bwt_table(struct bwt_table *bwt_table, struct suffix_array *sa, struct suffix_array *rsa,
                    struct remap_table *remap_table);
/* stralg/bwt.h:98-100 (bwt.c:109-118) */
struct bwt_table *alloc_bwt_table(struct suffix_array *sa, struct suffix_array *rsa,
                                  struct remap_table *remap_table);
/* stralg/bwt.h:110-139 (bwt.c:91-132) */
void dealloc_bwt_table(struct bwt_table *bwt_table);
void free_bwt_table(struct bwt_table *bwt_table);
void completely_dealloc_bwt_table(struct bwt_table *bwt_table);
void completely_free_bwt_table(struct bwt_table *bwt_table);
/* stralg/bwt.h:156-160 (bwt.c:134-161) */
struct bwt_table *build_complete_table(const uint8_t *string, bool include_reverse);

/* stralg/bwt.h:168-230 (bwt.c:164-223): exact FM-index search over host tables, one pattern at a time
 * (the batched device form is sx_bwt_exact_search_dev in stralg_amd.h) */
struct bwt_exact_match_iter {
    const struct suffix_array *sa;
    uint32_t L;
    int64_t i;
    uint32_t R;
};
struct bwt_exact_match {
    uint32_t pos;
};
void init_bwt_exact_match_iter(struct bwt_exact_match_iter *iter, struct bwt_table *bwt_table,
                               const uint8_t *remapped_pattern);
bool next_bwt_exact_match_iter(struct bwt_exact_match_iter *iter, struct bwt_exact_match *match);
void dealloc_bwt_exact_match_iter(struct bwt_exact_match_iter *iter);

/* ---- index serialisation: the reference's file format (stralg/serialise.h:8-23, string_utils.h,
 * suffix_array.h, remap.h, bwt.h write_ / read_ pairs) ------------------------------------------- */
/* (every FILE form has its _fname sibling: string_utils.h:48-71, suffix_array.h:109-126, remap.h:89-104, bwt.h:337-354) */
void write_string_len(FILE *f, const uint8_t *str, uint32_t len); /* string_utils.c:48-52 */
void write_string_len_fname(const char *fname, const uint8_t *str, uint32_t len);
void write_string_fname(const char *fname, const uint8_t *str);
uint8_t *read_string_len_fname(const char *fname, uint32_t *len);
uint8_t *read_string_fname(const char *fname);
void write_suffix_array_fname(const char *fname, const struct suffix_array *sa);
struct suffix_array *read_suffix_array_fname(const char *fname, uint8_t *string);
void write_remap_table_fname(const char *fname, const struct remap_table *table);
struct remap_table *read_remap_table_fname(const char *fname);
void write_bwt_table_fname(const char *fname, const struct bwt_table *bwt_table);
struct bwt_table *read_bwt_table_fname(const char *fname, struct suffix_array *sa, struct remap_table *remap_table);
void write_string(FILE *f, const uint8_t *str);                   /* string_utils.c:62-66: length includes the NUL */
uint8_t *read_string_len(FILE *f, uint32_t *len);                 /* string_utils.c:73-82 */
uint8_t *read_string(FILE *f);
void write_suffix_array(FILE *f, const struct suffix_array *sa);  /* suffix_array.c:238-241 */
struct suffix_array *read_suffix_array(FILE *f, uint8_t *string); /* suffix_array.c:250-258 */
void write_remap_table(FILE *f, const struct remap_table *table); /* remap.c:168-173 */
struct remap_table *read_remap_table(FILE *f);                    /* remap.c:184-191 */
void write_bwt_table(FILE *f, const struct bwt_table *bwt_table); /* bwt.c:425-441 */
struct bwt_table *read_bwt_table(FILE *f, struct suffix_array *sa, struct remap_table *remap_table); /* bwt.c:453-492 */
void write_complete_bwt_info(FILE *f, const struct bwt_table *bwt_table); /* serialise.c:7-18 */
void write_complete_bwt_info_fname(const char *fname, const struct bwt_table *bwt_table);
struct bwt_table *read_complete_bwt_info(FILE *f);                /* serialise.c:29-39 */
struct bwt_table *read_complete_bwt_info_fname(const char *fname);

/* ---- FASTA records (bioinf/fasta.h:10-49, stralg/error.h:7-21) --------------------------------
 * load_fasta_records reads the file and packs it on the GPU (sx_fasta_pack: the reference's in-place
 * packing loop as scan + compaction); names and sequences point into one buffer owned by the records
 * object, the iterator walks the records in reverse file order (fasta.c:131-134), sequences keep
 * their original symbols (build_complete_table remaps). */
enum error_codes {
    NO_ERROR,
    CANNOT_OPEN_FILE,
    MALFORMED_FILE,
    SUFFIX_ARRAYS_DIFFER,
    REMAP_TABLES_DIFFER,
    BWT_TABLES_DIFFER,
    MALFORMED_CIGAR
};
struct fasta_records;
struct fasta_record_impl;
struct fasta_record {
    const char *name;
    const uint8_t *seq;
    uint32_t seq_len;
};
struct fasta_iter {
    struct fasta_record_impl *rec;
};
struct fasta_records *load_fasta_records(const char *fname, enum error_codes *err);
void free_fasta_records(struct fasta_records *file);
uint32_t number_of_fasta_records(struct fasta_records *records);
bool lookup_fasta_record_by_name(struct fasta_records *file, const char *name, struct fasta_record *record);
void init_fasta_iter(struct fasta_iter *iter, struct fasta_records *file);
bool next_fasta_record(struct fasta_iter *iter, struct fasta_record *rec);
void dealloc_fasta_iter(struct fasta_iter *iter);

/* ---- additions (not in the reference) --------------------------------------- */
/* GPU used by the calling thread's constructors (default: $STRALG_AMD_DEVICE or 0).
 * One context per host thread, so N threads can farm records over N GPUs
 * (tools/readmappers/bwt_readmapper/bwt_readmapper.c:54-62 is the per-record loop). */
int stralg_amd_set_device(int device);
/* release the calling thread's context and its cached device memory (a thread that exits without this call releases
 * them too: the context hangs on a pthread key whose destructor runs at thread exit) */
void stralg_amd_release(void);
/* device contexts alive in this process */
int stralg_amd_live_contexts(void);
/* Host memory kept between calls.  Result arrays of 64 MiB and more that are freed through this library's free_* /
 * dealloc_* functions go to a cache of the calling thread (at most 8 blocks) and come back to its next build instead
 * of being unmapped and first-touched again.  All threads' caches together hold at most $STRALG_AMD_HOST_CACHE_GIB (default
 * 64, never more than half of physical memory; 0: no cache); a block that does not fit is free()d.  A thread's cached
 * blocks stay resident until it calls stralg_amd_release() or exits -- the reference's API has no such call, so a
 * long-lived caller keeps up to the cap as RSS after its last record.  This function: bytes cached by all threads now. */
size_t stralg_amd_host_cache_bytes(void);
/* strlen(string) as build_complete_table takes it for long records -- one walk on the calling thread, aligned 32-byte
 * loads, never a byte behind the terminator's own 32-byte block (as strlen) -- and the byte values the string holds
 * (present[256], filled when *have_letters comes back 1: strings of 4 MiB and more) */
size_t stralg_amd_strlen_and_letters(const uint8_t *string, uint8_t *present, int *have_letters);
/* Build tables for `count` independent strings over the listed devices by host threads pinned to their GPU's NUMA
 * node -- one a device for long records, up to four for records too short to fill a GPU
 * (stralg_amd_farm_workers_per_device); strings are dealt longest first to the least loaded worker (LPT by length).
 * out[k] receives build_complete_table(strings[k], ...).  Returns 0, the number of records that could not be built
 * (more letters than a remap table holds, no memory on the host or the device: out[k] == NULL for each, a line on
 * stderr, the other records are built all the same), or a negative value when the farm itself could not be set up. */
int stralg_amd_build_tables_batch(const uint8_t *const *strings, size_t count, bool include_reverse,
                                  const int *devices, int n_devices, struct bwt_table **out);
/* workers (contexts, host threads) the farm runs on each device for records of these lengths; $STRALG_AMD_FARM_WORKERS
 * overrides */
int stralg_amd_farm_workers_per_device(const size_t *lengths, size_t count, int n_devices);
/* the assignment the farm uses: assignment[k] = lane (0 .. lanes-1) of record k */
int stralg_amd_lpt_assign(const size_t *lengths, size_t count, int lanes, int *assignment);
/* pin the calling thread to the CPUs of `device`'s NUMA node; returns the node, or -1 when it is unknown */
int stralg_amd_bind_thread_to_device(int device);
/* build_complete_table(string, include_reverse) followed by write_complete_bwt_info(f, table), byte for byte,
 * but the suffix array and the O tables stream from the GPU into the file in 32 MiB chunks: no host copy of
 * the tables (20 GiB per GiB of DNA).  Returns 0, or a negative / HIP error code. */
int stralg_amd_write_complete_bwt_info_stream(FILE *f, const uint8_t *string, bool include_reverse);
/* The loop of bwt_readmapper.c:54-62 over a whole FASTA file: out[k] = build_complete_table of the k-th
 * record in ITERATION order (reverse file order), records farmed over the devices; returns the number of
 * records (out needs number_of_fasta_records(records) slots), or a negative value on failure.  A record that could not be
 * built (more letters than a remap table holds, no memory) leaves out[k] == NULL: check every entry, or call the _ex form,
 * which also stores the number of such records in *n_failed (may be NULL). */
int stralg_amd_fasta_tables_batch_ex(struct fasta_records *records, bool include_reverse, const int *devices,
                                     int n_devices, struct bwt_table **out, size_t *n_failed);
int stralg_amd_fasta_tables_batch(struct fasta_records *records, bool include_reverse, const int *devices,
                                  int n_devices, struct bwt_table **out);

#ifdef __cplusplus
}
#endif
#endif
