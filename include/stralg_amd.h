/*
 * stralg_amd.h -- C-ABI of the MI355X suffix-array / BWT-table construction
 * path (libstralg_amd.so).  Plain pointers and sizes only.
 *
 * These are the device-side entry points that stralg's own constructors bind
 * (include/stralg_compat.h declares the reference-named wrappers on top):
 *
 *   sx_sa_build      replaces the body of  sa_is_construction      stralg/sa_is.c:466-509
 *                                          sa_is_mem_construction  stralg/sa_is_mem.c:471-494
 *                                          skew_sa_construction    stralg/skew.c:388-395
 *   sx_bwt_tables    replaces the C/O/RO loops of init_bwt_table   stralg/bwt.c:35-88
 *
 * Every function returns 0 on success or a non-zero code (HIP error number,
 * or one of SX_E_*); sx_last_error() gives the text.  There is no CPU
 * fallback: without a usable GPU the calls fail.
 */
#ifndef STRALG_AMD_H
#define STRALG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sx_ctx sx_ctx;

enum {
    SX_OK = 0,
    SX_E_ARG = -1,     /* malformed argument (symbol >= alphabet_size, interior 0, n too large) */
    SX_E_NOMEM = -2,   /* host allocation failed */
    SX_E_INTERNAL = -3, /* a device-side invariant did not hold */
    SX_E_MALFORMED = -4 /* a FASTA image that ends inside a header line (bioinf/fasta.c:121-124 MALFORMED_FILE) */
};

/* Kernel classes for the in-library HIP-event profiler (bench.py roofline). */
enum {
    SX_KC_CLASSIFY = 0,   /* S/L types, LMS flags, bucket histograms      sa_is.c:134-174 */
    SX_KC_SAMPLES,        /* sample (LMS + cut) flags and compaction                       */
    SX_KC_KEYS,           /* LMS-substring pieces / prefixes -> 64-bit keys sa_is.c:265-292; a sort's first pass when it
                             computes its keys from the text itself (no key kernel) */
    SX_KC_RADIX_HIST,     /* radix sort: per-tile digit histogram                          */
    SX_KC_RADIX_SCATTER,  /* radix sort: stable scatter                                    */
    SX_KC_SCAN,           /* device-wide scans / compactions                               */
    SX_KC_NAMES,          /* names + reduced string                       sa_is.c:295-336 */
    SX_KC_DOUBLING,       /* reduced-string suffix sort (elementwise steps)                */
    SX_KC_INDUCE_GATHER,  /* induce: gather text[SA[i]-1] + per-tile bucket histogram      */
    SX_KC_INDUCE_SCAN,    /* induce: per-bucket offsets                                    */
    SX_KC_INDUCE_SCATTER, /* induce: stable scatter to bucket cursors     sa_is.c:220-263 */
    SX_KC_INDUCE_CHAIN,   /* induce: small rounds, one chained launch (count + look-back + scatter) */
    SX_KC_BWT_GATHER,     /* bwt[i] = text[SA[i]-1] + per-tile symbol counts bwt.c:13-20  */
    SX_KC_OTABLE,         /* O-table rows                                 bwt.c:47-65     */
    SX_KC_MISC,
    SX_KC_FASTA,          /* FASTA image -> packed records                bioinf/fasta.c:92-135 */
    SX_KC_REMAP,          /* presence bits + table lookup                 remap.c:8-31,102-114  */
    SX_KC_LCP,            /* inverse + LCP                                suffix_array.c:53-85  */
    SX_KC_SEARCH,         /* batched exact BWT search                     bwt.c:164-199         */
    SX_KC_LOCAL_SORT,     /* hybrid LMS sort: sub-buckets ordered in LDS, ties listed          */
    SX_KC_COUNT
};

typedef struct sx_kernel_stat {
    uint64_t launches;
    double ms;          /* sum of HIP-event durations */
    uint64_t alg_bytes; /* sum of algorithmic bytes (per kernel) */
} sx_kernel_stat;

typedef struct sx_build_stats {
    uint64_t n;              /* symbols without the sentinel */
    uint64_t n_lms;          /* LMS positions incl. the sentinel */
    uint64_t n_samples;      /* LMS positions + cut points = reduced string length */
    uint64_t n_names;        /* distinct piece names */
    uint32_t key_bits;       /* bits per symbol in a piece key */
    uint32_t key_slots;      /* symbols per piece key */
    uint32_t doubling_rounds;
    uint32_t induce_rounds;  /* multisplit rounds over both passes */
    uint32_t sort_passes;    /* radix passes, all sorts */
    uint32_t lms_path;       /* 1: prefix-key LMS sort resolved everything, 2: general path, 3: direct sort of all suffixes */
    uint32_t sort_local;     /* bit 0: the prefix-key sort finished in LDS (hybrid: HBM passes on the top 24 key bits only);
                                bit 1: some workgroup of it met crowded bins and took stable passes; bit 2: HBM passes on the top 32 bits (four);
                                bit 3: its first HBM pass computed the keys from the text (no key kernel) */
    uint32_t refine_tiers;   /* tie refinement of the prefix-key sort: bit 0: some round ordered groups of 9 .. 2048 members in
                                LDS; bit 1: some round sent the members of longer groups through radix sorts; prefix doubling of the
                                general path: bit 2: some round ordered small groups by one wave each, bit 3: some round sent
                                members through radix sorts */
    double ms_total;         /* wall time of the last build on the device stream */
    uint32_t induce_redo;    /* buckets of the induced-sort passes whose rounds the queued launches did not finish (runs longer than
                                the tail kernel's steps reach, more entries alive than it holds): the host carried them on */
    uint32_t long_runs;      /* the classification saw a run that fills a 4096-symbol tile: the passes are attended from the start */
    uint32_t recursion_levels; /* reduced strings over a byte alphabet that were sorted by the pipeline itself, one below the other */
    uint32_t sample_tied_permille; /* 0: no sample was looked at; else 1 + the tied share (per mille) of the sampled suffixes under
                                      the longest prefix key: from 300 on the prefix-key sort is not attempted */
    uint32_t long_subbuckets; /* hybrid prefix-key sort: sub-buckets too long for a workgroup's LDS (repeat families, AT-rich
                                 prefixes) that were ordered by HBM passes of their own */
} sx_build_stats;

/* ---- context ------------------------------------------------------------ */
int sx_device_count(void);
/* NUMA node of the device's PCI function (/sys/bus/pci/devices/<bus id>/numa_node), or -1 when unknown */
int sx_device_numa_node(int device);
int sx_ctx_create(int device, sx_ctx **out);
void sx_ctx_destroy(sx_ctx *ctx);
/* contexts alive in this process (created minus destroyed) */
int sx_ctx_live_count(void);
const char *sx_last_error(const sx_ctx *ctx);
/* drop cached workspace (it is otherwise kept between calls) */
void sx_ctx_trim(sx_ctx *ctx);
/* behaviour switches (testing / measurement) */
enum {
    SX_FLAG_FORCE_GENERAL_PATH = 1, /* skip the prefix-key LMS sort: always pieces + names + prefix doubling */
    SX_FLAG_CHAIN_MAX_ENTRIES = 2,  /* induce rounds up to this many entries use the single chained launch */
    SX_FLAG_NO_DIRECT_SORT = 3,     /* wide alphabets: never sort all suffixes by prefix directly, always LMS sort + induction */
    SX_FLAG_PREFIX_SYMBOLS = 4,     /* first attempt of the prefix-key sort takes this many symbols (0: by the text's size) */
    SX_FLAG_RADIX_DIGIT_BITS = 5,   /* digit width of the LSD radix passes: 8 (default), 9 or 10 */
    SX_FLAG_SORT_MODE = 6,          /* prefix-key sort: 0 choose, 1 LSD passes only (tie refinement too: no group is ordered in
                                       LDS), 2 hybrid (HBM passes on the top 24 key bits + sub-buckets ordered in LDS) whenever the
                                       key shape allows it, whatever the size, 3 the same with the top 32 bits */
    SX_FLAG_INDUCE_BATCH_OFF = 7,   /* induced-sort passes over at most 8 buckets: 1 = every self round of a bucket is a launch of
                                       its own (no eight-rounds-at-a-time form) */
    SX_FLAG_INDUCE_BATCH_MIN = 8,   /* ranges longer than this many entries take the eight-rounds-at-a-time form (negative: the
                                       default, what the one-workgroup tail kernel holds; tests set 0) */
    SX_FLAG_INDUCE_ATTENDED = 9     /* 0 (default) = the buckets of an induced-sort pass are queued one behind the other; a bucket
                                       whose rounds the tail kernel could not finish leaves word, the launches behind it do nothing,
                                       and the host carries that bucket on before it queues the rest; 1 = attended: the host reads
                                       every bucket's last range back before it queues the next bucket (rounds 1 and 2) */
    ,SX_FLAG_COPY_TEXT_FIRST = 10   /* 1 = the build's padded copy of the text is made by a device copy before the classification
                                       (rounds 1 and 2); 0 = the classification writes it while it reads the caller's text */
    ,SX_FLAG_RECURSE_MIN = 11      /* a reduced string of at most 255 names and at least this many symbols is sorted by the whole
                                       pipeline again (in a child context) instead of by prefix doubling; negative: the default
                                       (2^20); tests set small values */
    ,SX_FLAG_SAMPLE_MIN = 12       /* texts of more than 8 symbols and at least this many suffixes get a look at a sample before a
                                       prefix-key sort (negative: the default, 2^20; tests set small values) */
    ,SX_FLAG_INDUCE_NO_HOIST = 13  /* texts of more than 8 symbols: 1 = every bucket's LMS seeds (L pass) and L-type entries (S pass)
                                       are scanned by launches of the bucket's own, as in rounds 1 - 3; 0 (default) = all buckets'
                                       at once, up front, placed by the text's bigram counts */
    ,SX_FLAG_TEXT_KEYS_OFF = 14    /* the direct sort of all suffixes and the LMS sort of four-letter texts: 1 = a key kernel writes
                                       the keys before the first radix pass (rounds 1 - 3); 0 (default) = the first pass computes them
                                       from the text */
    ,SX_FLAG_LONG_SUBBUCKETS_OFF = 15 /* hybrid prefix-key sort: 1 = a sub-bucket too long for a workgroup makes the whole sort fall
                                       back to plain passes, and texts with skewed symbol counts do not try it (rounds 1 - 3); 0
                                       (default) = such sub-buckets are listed and ordered by HBM passes of their own */
    ,SX_FLAG_SMALL_DIRECT_MAX = 16 /* texts of at most 16 symbols and at most this many suffixes are sorted directly (all suffixes by
                                       prefix key, as wide alphabets are: a third of the launches of classification + LMS sort +
                                       induced passes, which is what a short record's build consists of); 0 = never; negative: the
                                       default (2^24, 2^25, 2^27 suffixes for at most 4, 7, 15 letters) */
    ,SX_FLAG_LOCAL_SORT_LEAN_OFF = 17 /* hybrid prefix-key sort, the step that orders the sub-buckets in LDS: 1 = every workgroup
                                       takes the kernel of rounds 3 and 4 (pairs through LDS, stable passes where equal keys
                                       crowd a bin); 0 (default) = the lean kernel of round 5, which leaves only the workgroups it
                                       cannot finish to that one */
};
int sx_ctx_set_flag(sx_ctx *ctx, int flag, int value);

/* ---- suffix array -------------------------------------------------------- */
/* Host buffers.  text[0..n) holds symbols in [1, alphabet_size), alphabet_size
 * <= 256; sa_out receives n+1 entries (sa_out[0] == n, the sentinel suffix).
 * Unlike sort_SA's shortcut (sa_is.c:423-428) the result is the true suffix
 * array for any alphabet_size that bounds the symbols. */
int sx_sa_build(sx_ctx *ctx, const uint8_t *text, uint64_t n, uint32_t alphabet_size,
                uint32_t *sa_out);
/* Device buffers (inputs resident in HBM); d_text has n bytes, d_sa_out n+1 entries. */
int sx_sa_build_dev(sx_ctx *ctx, const uint8_t *d_text, uint64_t n, uint32_t alphabet_size,
                    uint32_t *d_sa_out);

/* Suffix array and BWT in one build: the induced-sort passes carry text[SA[i]-1]
 * with every entry, so the BWT costs no extra gather.  d_bwt_out has n+1 bytes. */
int sx_sa_bwt_build_dev(sx_ctx *ctx, const uint8_t *d_text, uint64_t n, uint32_t alphabet_size,
                        uint32_t *d_sa_out, uint8_t *d_bwt_out);

/* ---- BWT tables ------------------------------------------------------------ */
/* text[0..N-1) symbols in [1, sigma) (N = n+1 counts the sentinel), sa[N].
 * c_out[sigma]; o_out[(N+1)*sigma] position-major: o_out[i*sigma + a] = O(a,i).
 * o_out may be NULL (C table only).  sigma <= 128 as in stralg/remap.h:14-18. */
int sx_bwt_tables(sx_ctx *ctx, const uint8_t *text, const uint32_t *sa, uint64_t N,
                  uint32_t sigma, uint32_t *c_out, uint32_t *o_out);
/* Device buffers; d_bwt_out (N bytes) is optional. */
int sx_bwt_tables_dev(sx_ctx *ctx, const uint8_t *d_text, const uint32_t *d_sa, uint64_t N,
                      uint32_t sigma, uint32_t *d_c_out, uint32_t *d_o_out, uint8_t *d_bwt_out);

/* C/O tables from a BWT already on the device (pairs with sx_sa_bwt_build_dev). */
int sx_bwt_tables_from_bwt_dev(sx_ctx *ctx, const uint8_t *d_bwt, uint64_t N, uint32_t sigma,
                               uint32_t *d_c_out, uint32_t *d_o_out);
/* build_complete_table's device work in one call (stralg/bwt.c:134-161): host text ->
 * suffix array (sa_out, n+1 entries, may be NULL), C table, O table (may be NULL). */
int sx_build_tables(sx_ctx *ctx, const uint8_t *text, uint64_t n, uint32_t sigma, uint32_t *sa_out,
                    uint32_t *c_out, uint32_t *o_out);

/* ---- consumers of a resident suffix array / table (SURVEY.md section 8f "next") ------------ */
/* stralg/suffix_array.c:53-60 compute_inverse: inv[sa[i]] = i. */
int sx_sa_inverse_dev(sx_ctx *ctx, const uint32_t *d_sa, uint64_t N, uint32_t *d_inv_out);
/* stralg/suffix_array.c:62-85 compute_lcp: lcp[0] = 0, lcp[j] = lcp(suffix sa[j-1], suffix sa[j]).
 * d_text has N-1 bytes; d_inv_out (N entries) is optional and receives the inverse. */
int sx_sa_lcp_dev(sx_ctx *ctx, const uint8_t *d_text, const uint32_t *d_sa, uint64_t N, uint32_t *d_inv_out,
                  uint32_t *d_lcp_out);
/* host buffers; inv_out or lcp_out may be NULL (not both) */
int sx_sa_inverse_lcp(sx_ctx *ctx, const uint8_t *text, const uint32_t *sa, uint64_t N, uint32_t *inv_out,
                      uint32_t *lcp_out);
/* stralg/bwt.c:164-199 init_bwt_exact_match_iter for `count` patterns at once: pattern q is
 * d_patterns[d_offsets[q] .. d_offsets[q+1]) (remapped symbols); the matches of q are
 * sa[l_out[q] .. r_out[q]) (empty when l_out[q] >= r_out[q]).  Tables as sx_bwt_tables_dev writes them. */
int sx_bwt_exact_search_dev(sx_ctx *ctx, const uint32_t *d_c_table, const uint32_t *d_o_table, uint64_t N,
                            uint32_t sigma, const uint8_t *d_patterns, const uint32_t *d_offsets, uint32_t count,
                            uint32_t *d_l_out, uint32_t *d_r_out);

/* ---- streaming download (serialisation without a host copy of the tables) ---- */
/* sink(user, section, data, bytes): consecutive chunks of one section after the other; data is only valid
 * during the call; a non-zero return aborts the build. */
typedef int (*sx_sink_fn)(void *user, int section, const void *data, size_t bytes);
enum { SX_SECTION_SA = 0, SX_SECTION_C = 1, SX_SECTION_O = 2 };
/* sx_build_tables, but the suffix array (when want_sa), the C table and the O table leave the device through
 * `sink` in 32 MiB chunks from pinned staging memory, in this order (the order of stralg/serialise.c:7-18 around
 * the remap table); the copy of a chunk overlaps the sink's work on the previous one. */
int sx_build_tables_stream(sx_ctx *ctx, const uint8_t *text, uint64_t n, uint32_t sigma, int want_sa, sx_sink_fn sink,
                           void *user);

/* ---- FASTA ingest and remap on the device ---------------- */
/* bioinf/fasta.c:92-135 load_fasta_records' packing of a file image in device memory into
 * "name\0sequence\0name\0sequence\0..." (file order; the reference's record list is the reverse).
 * d_packed_out: file_len + 1 bytes.  d_term_out (optional, term_cap entries): positions of the
 * terminators in the packed image, so record r has its name at (r ? term[2r-1] + 1 : 0), its
 * sequence at term[2r] + 1 and seq_len = term[2r+1] - term[2r] - 1.  file_len < 2^31 - 1.
 * Returns SX_E_MALFORMED where the reference reports MALFORMED_FILE. */
int sx_fasta_pack_dev(sx_ctx *ctx, const uint8_t *d_file, uint64_t file_len, uint8_t *d_packed_out,
                      uint64_t *packed_len_out, uint32_t *d_term_out, uint64_t term_cap, uint32_t *n_records_out);
/* the same with host buffers (staged through the context) */
int sx_fasta_pack(sx_ctx *ctx, const uint8_t *file, uint64_t file_len, uint8_t *packed_out, uint64_t *packed_len_out,
                  uint32_t *term_out, uint64_t term_cap, uint32_t *n_records_out);
/* stralg/remap.c:8-31,102-114 build_remap_table + remap: d_out[0..n) = dense order-preserving codes 1..k of
 * d_in, d_out[n] = 0; table_out (host, 256 entries, optional): code of every byte value, -1 for absent ones;
 * *alphabet_size_out = k + 1.  Fails when more than 127 distinct symbols occur (remap.h:14-18). */
int sx_remap_dev(sx_ctx *ctx, const uint8_t *d_in, uint64_t n, uint8_t *d_out, int16_t *table_out,
                 uint32_t *alphabet_size_out);
/* stralg/bwt.c:147-151: the reversed copy of a remapped string that build_complete_table sorts for the RO table:
 * d_out[i] = d_in[n - 1 - i], d_out[n] = 0 (n + 1 bytes; the buffers must not overlap) */
int sx_reverse_dev(sx_ctx *ctx, const uint8_t *d_in, uint64_t n, uint8_t *d_out);

/* ---- measurement ------------------------------------------------------------ */
int sx_profile_enable(sx_ctx *ctx, int on);       /* bracket every launch with HIP events */
int sx_profile_only(sx_ctx *ctx, int kclass);     /* ... only launches of this class (kclass < 0: all): two event records per
                                                     launch cost ~5 % on a build of 300 short launches */
int sx_profile_reset(sx_ctx *ctx);
int sx_profile_read(sx_ctx *ctx, sx_kernel_stat *out /* SX_KC_COUNT entries */);
const char *sx_kernel_class_name(int kclass);
int sx_last_stats(const sx_ctx *ctx, sx_build_stats *out);

/* Synthetic input on the device: symbol i = 1 + (splitmix64(seed, i) >> 33) % (sigma - 1)
 * (same stream as oracle_synth / stralg_amd.synth). */
int sx_synth_dev(sx_ctx *ctx, uint8_t *d_out, uint64_t n, uint32_t sigma, uint64_t seed);

/* The box's memory ceiling (measurement aid; SURVEY.md section 8d "confirm on the box"): four streaming shapes over two
 * device buffers of `bytes` each (16-byte aligned; use far more than the 256 MB last-level cache), HIP-event timed on the
 * context's stream, best of `reps` after a warm-up.  out_GBps[0] read (16 B a lane), [1] fill, [2] copy (bytes counted both
 * ways), [3] four-way split of 4-byte entries (4 B in + 4 B out an entry: the store shape of the induced-sort scatters with
 * no ranking work).  d_b is overwritten. */
int sx_membw_probe(sx_ctx *ctx, void *d_a, void *d_b, uint64_t bytes, int reps, double *out_GBps /* 4 */);

/* ---- primitives, exported for the kernel-level tests ---------------------- */
/* stable LSD radix sort of (u64 key, u32 value) pairs on bits [begin_bit, end_bit);
 * all four device buffers hold n entries; *result_in_b tells where the output is. */
int sx_prim_sort_pairs_dev(sx_ctx *ctx, uint64_t *d_keys_a, uint32_t *d_vals_a, uint64_t *d_keys_b,
                           uint32_t *d_vals_b, uint64_t n, int begin_bit, int end_bit,
                           int *result_in_b);
/* exclusive prefix sum of n u32; d_total (optional) receives the grand total */
int sx_prim_exclusive_sum_dev(sx_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint64_t n,
                              uint32_t *d_total);
/* S/L classification products: LMS flags as one byte per position (n+1) and the
 * three per-symbol histograms (256 entries each): all symbols incl. sentinel,
 * L-type symbols, LMS symbols. */
int sx_prim_classify_dev(sx_ctx *ctx, const uint8_t *d_text, uint64_t n, uint8_t *d_lms_flags,
                         uint32_t *d_hist_all, uint32_t *d_hist_l, uint32_t *d_hist_lms);

#ifdef __cplusplus
}
#endif
#endif
