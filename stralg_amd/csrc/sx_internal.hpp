// sx_internal.hpp -- stage interfaces between the translation units.
#pragma once
#include "sx_common.hpp"

// ---- sx_radix.hip
// Keys that are a function of the text: pair i of a sort of ALL suffixes has the key of text[i .. i + C) in base `base`
// (most significant symbol first) and, with wnd, the symbol in front of position i as a one-symbol window above bit
// kbits (sx_window.hpp).  A sort given such a description computes the keys in its first pass instead of reading them
// (the direct sort of wide alphabets: no key kernel, no 8 bytes a suffix written and read back).
namespace sx { struct sx_lmskey; }
struct sx_textkey {
    const uint8_t *T; // the build's padded copy of the text (readable 16 bytes beyond any position)
    uint32_t base, C; // C <= 12
    uint32_t pow3, powR; // base^3, base^(C mod 3)
    uint32_t kbits, wnd;
};
int sx_sort_pairs(sx_ctx *ctx, uint64_t *ka, uint32_t *va, uint64_t *kb, uint32_t *vb, uint64_t n,
                  int begin_bit, int end_bit, int *result_in_b, bool values_are_indices = false,
                  bool first_digits_ready = false, int digit_bits = 0 /* 8, 9, 10; 0: the context's choice */,
                  const sx_textkey *text_keys = nullptr /* the first pass computes the keys (ka is not read; values = indices) */,
                  const sx::sx_lmskey *lms_keys = nullptr /* the first pass lists the LMS suffixes and computes their keys (sx_lmskey.hpp;
                                                             ka, va are not read, ka holds that pass's tile table) */);
// whether a sort of m LMS suffixes of a text of cls_tiles classification tiles can take that first pass for keys of this shape
bool sx_sort_lms_keys_applies(uint64_t m, uint32_t cls_tiles, uint32_t shape);
// where the key generator of a sort of n pairs may leave the first pass's digit of every key
// ((key >> begin_bit) & (2^digit_bits - 1); one byte each for 8-bit digits, two for wider ones): the first
// histogram then reads these instead of the keys (first_digits_ready)
void *sx_sort_digit_buffer(sx_ctx *ctx, uint64_t n, int digit_bits = 8);
// digit width of sorts that do not ask for one (8 unless SX_FLAG_RADIX_DIGIT_BITS says otherwise)
int sx_sort_digit_bits(const sx_ctx *ctx);

// ---- sx_classify.hip
constexpr int kClsPerThread = 16;                     // text positions per thread
constexpr int kClsTile = 256 * kClsPerThread;         // text positions per workgroup

struct sx_text_info {
    const uint8_t *T;   // device text: n symbols, T[n] == 0, zero padding to the tile end + 32
    uint64_t n, N;      // N = n + 1 positions
    uint32_t ntiles;    // classification tiles
    uint16_t *lmsbits;  // one bit per position, 16 positions per entry
    uint16_t *sampbits; // sample (LMS or cut) bits, same layout
    uint32_t *tile_u32; // 5 * ntiles scratch: lms count, last lms(+1), prev lms(+1), sample count, sample offset
    uint8_t *tile_first; // per tile: type of its first position (0 L, 1 S, 2 open)
    uint32_t *d_hist;   // device: 3 * 256: all symbols, L-type symbols, LMS symbols
    uint32_t *d_scalar; // device: a few u32 results (totals)
    uint32_t h_all[256], h_l[256], h_lms[256];
    uint32_t open_tiles; // classification tiles made of one symbol whose run goes on: runs of 4096 symbols and more
    uint32_t maxc;      // largest symbol present
    uint64_t m;         // LMS positions incl. the sentinel
    uint64_t M;         // samples
};

size_t sx_text_scratch_bytes(uint64_t n);
// symbol counts of T[0..n) (the sentinel at n is not counted); d_scratch256: 256 u32 of device scratch
int sx_symbol_histogram(sx_ctx *ctx, const uint8_t *T, uint64_t n, uint32_t *d_scratch256, uint32_t h_out[256]);
// carve the per-text scratch out of `arena`, run classification, read the histograms back.  src / src_tiles: the first
// src_tiles classification tiles of T have not been copied from the caller's text `src` yet (16-byte aligned): the
// classification reads them there and writes them into T as it goes (sx_classify.hip: cls_types_kernel)
int sx_classify(sx_ctx *ctx, uint8_t *T, uint64_t n, sx_arena &arena, sx_text_info &ti, const uint8_t *src = nullptr,
                uint32_t src_tiles = 0);
// sample flags for piece width W (symbols between consecutive samples <= W); sets ti.M
int sx_sample_flags(sx_ctx *ctx, sx_text_info &ti, uint32_t W);
// compaction of the sample positions; pos[M], is_lms[M]
int sx_sample_write(sx_ctx *ctx, const sx_text_info &ti, uint32_t *pos, uint8_t *is_lms);
int sx_piece_keys(sx_ctx *ctx, const sx_text_info &ti, const uint32_t *pos, const uint8_t *is_lms,
                  uint32_t bits, uint32_t slots, uint32_t lenbits, uint64_t *keys, uint32_t *vals);

// ---- sx_reduce.hip
struct sx_reduce_bufs {
    uint64_t *ka, *kb;      // M keys each
    uint32_t *va, *vb;      // M values each
    uint32_t *R;            // reduced string (names), M
    uint32_t *rank;         // M
    uint32_t *sa_r;         // suffix array of the reduced string, M
    uint32_t *pos_a, *pos_b; // active positions, M each
    uint32_t *gid;          // M
    uint32_t *sub_t;        // M: list slots of the members a doubling round orders by radix sorts
    uint8_t *head_a, *head_b; // M each
    uint32_t *d_scalar;     // >= 4 u32
    uint32_t *head_bins;    // 1024 u32: partial counts of the groups a doubling round leaves
};
// names from the sorted piece keys; n_names out.  keys/vals sorted in (ks, vs).
int sx_name_pieces(sx_ctx *ctx, const uint64_t *ks, const uint32_t *vs, uint64_t M, sx_reduce_bufs &rb,
                   uint64_t *n_names);
// suffix array of R[0..M) (R[M-1] == 0 unique minimum) by prefix doubling -> rb.sa_r
int sx_reduced_suffix_sort(sx_ctx *ctx, uint64_t M, uint64_t n_names, sx_reduce_bufs &rb);
// sorted LMS suffix positions from the reduced suffix array
int sx_sorted_lms(sx_ctx *ctx, const uint32_t *sa_r, const uint32_t *pos, const uint8_t *is_lms, uint64_t M,
                  uint64_t m, uint32_t *sorted_lms, uint32_t *d_total);

// the same with every suffix's window (sx_window.hpp; what sx_induce takes as seed_windows) from one gather; buf_a, buf_b:
// scratch of M x 8 bytes each (M x 16 for texts of more than 16 symbols, whose windows are 64-bit words); *seed_windows
// lies in one of them
int sx_sorted_lms_windows(sx_ctx *ctx, const sx_text_info &ti, const uint32_t *sa_r, const uint32_t *pos, const uint8_t *is_lms, uint64_t M,
                          uint64_t m, void *buf_a, void *buf_b, uint32_t *sorted_lms, const void **seed_windows, uint32_t *d_total);

// ---- sx_lmssort.hip
size_t sx_lms_prefix_bytes(uint64_t m);
int sx_bwt_from_seed_windows(sx_ctx *ctx, const uint32_t *seedw, uint64_t N, uint32_t maxc, uint8_t *bwt_out);
// tied share of a sample under the longest prefix key (a lower bound of the whole's); < 0: no look taken (at most 8 symbols,
// short texts)
int sx_prefix_ties_sampled(sx_ctx *ctx, const sx_text_info &ti, sx_arena am, bool all_suffixes, double *share);
int sx_sort_lms_by_prefix(sx_ctx *ctx, const sx_text_info &ti, sx_arena &am, const uint32_t **out,
                          const void **seed_windows, int *resolved, bool all_suffixes = false);

// ---- sx_localsort.hip: the hybrid sort's last step (sub-buckets of equal top key bits ordered in LDS)
// key bits that go through HBM passes: 24 (three 8-bit passes) or, when those leave sub-buckets too long for a workgroup
// (texts of 2 Gi symbols and more, skewed symbol frequencies), 32 (four)
bool sx_local_sort_applies(uint64_t m, int kbits, int top_bits);
uint32_t sx_local_sort_tiles(uint64_t m);
// (kin, vin): m pairs ordered by key bits [kbits - top_bits, kbits).  Writes the positions in key order to vout, the keys'
// payload bits (from kbits on) to seedw (optional), the members of groups of equal keys to (apos, ap, ahead)
// (at most cap), their number to d_total_and_fail[0]; d_total_and_fail[1] <- bit 0 when a sub-bucket did not fit a
// workgroup (the outputs are then unusable), bit 1 when some workgroup used stable passes (statistics).  tile_*: one u32 per workgroup (sx_local_sort_tiles); stage: m x 8
// bytes, stage_head: m bytes of scratch.
int sx_local_sort(sx_ctx *ctx, const uint64_t *kin, const uint32_t *vin, uint64_t m, int kbits, int top_bits, uint32_t *vout,
                  uint32_t *seedw, uint32_t *tile_start, uint32_t *tile_cnt, uint32_t *tile_off, uint2 *stage,
                  uint8_t *stage_head, uint32_t *apos, uint32_t *ap, uint8_t *ahead, uint32_t cap,
                  uint32_t *d_total_and_fail, uint32_t longest_expected = 0 /* the longest sub-bucket the caller expects; 0: not known */,
                  uint32_t *long_list = nullptr /* 3 * long_cap words: sub-buckets too long for a workgroup are listed (d_total_and_fail[2]
                                                   <- their number) instead of failing the sort; sx_long_subbuckets finishes them */,
                  uint32_t long_cap = 0, uint32_t res[3] = nullptr /* d_total_and_fail[0 .. 3), read back (round 5: the host looks at bit 2 of
                                                                     [1] -- workgroups that asked for the stable-pass kernel -- anyway) */,
                  bool crowded_expected = false /* skewed symbol counts: many workgroups would be left to the stable-pass kernel, which then
                                                   takes all of them at once */);
int sx_long_subbuckets(sx_ctx *ctx, const uint64_t *kin, const uint32_t *vin, uint64_t m, int kbits, int top_bits, uint32_t n_long,
                       uint32_t *long_list, uint32_t long_cap, uint64_t *ck_a, uint64_t *ck_b, uint32_t *cv_a, uint32_t *cv_b,
                       uint32_t max_pairs, uint32_t *vout, uint32_t *seedw, uint32_t *apos, uint32_t *ap, uint8_t *ahead, uint32_t cap,
                       uint32_t tied0, uint32_t *d_scalar2, uint32_t *tied_total, uint32_t *pairs, int *done);

// ---- sx_induce.hip
size_t sx_induce_scratch_bytes(uint64_t N, uint32_t sigma);
// seed_windows: optional device array (one window per sorted LMS suffix, layout of sx_window.hpp); seed_windows_u32: they
// are 32-bit words although the text's windows are 64-bit (the prefix-key sort's: fewer symbols, same layout)
int sx_induce(sx_ctx *ctx, const sx_text_info &ti, uint32_t sigma, const uint32_t *sorted_lms,
              const void *seed_windows, bool seed_windows_u32, uint32_t *SA, uint8_t *bwt_out, sx_arena &arena);

// ---- sx_extras.hip: out[targets[i]] = values[i] (null: i) for a permutation of [0, N), in two passes that keep the stores
// inside windows the L2 holds (half the time of the plain scatter from 2^23 entries on)
bool sx_scatter_permutation_applies(uint64_t N);
size_t sx_scatter_permutation_cursor_words();
int sx_scatter_permutation(sx_ctx *ctx, const uint32_t *targets, const uint32_t *values, uint64_t N, uint32_t *out, void *pairs_scratch,
                           uint32_t *cursor, uint32_t *bad, int kclass);

// ---- sx_build.hip / sx_bwt.hip (shared by the fused host entry point)
int sx_sa_build_impl(sx_ctx *ctx, const uint8_t *d_text, uint64_t n, uint32_t sigma, uint32_t *d_sa, uint8_t *d_bwt);
int sx_tables_from_bwt_impl(sx_ctx *ctx, const uint8_t *d_bwt, uint64_t N, uint32_t sigma, uint32_t *d_c, uint32_t *d_o);
