/*
 * oracle.h -- CPU restatement of stralg's SA-IS / BWT-table path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The shipped library (stralg_amd/csrc) never links, loads or calls it.
 *
 * Parity status: PINNED.  oracle/_ref/libstralg_ref.so (the unmodified
 * reference, compiled from /root/reference by oracle/Makefile) and every
 * golden vector the reference's own tests hold for this path are checked
 * against this restatement by tests/test_oracle.py; the outputs of the
 * reference run in the build container are committed under tests/golden/.
 *
 * Every function cites the reference file:line it follows (paths relative
 * to the reference checkout).
 */
#ifndef STRALG_AMD_ORACLE_H
#define STRALG_AMD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* stralg/sa_is.c:466-509 sa_is_construction.  text[0..n) holds symbols in
 * [1, sigma); the sentinel 0 is implicit at position n.  Writes the n+1
 * entries of the suffix array.  Returns 0, or -1 on allocation failure. */
int oracle_sa_is(const uint8_t *text, size_t n, uint32_t sigma, uint32_t *sa_out);

/* As above but with sort_SA's "sigma == n + 1" shortcut (sa_is.c:423-428)
 * disabled at the top level: gives the mathematically correct array for a
 * loose alphabet_size. */
int oracle_sa_is_strict(const uint8_t *text, size_t n, uint32_t sigma, uint32_t *sa_out);

/* stralg/suffix_array.c:32-48 qsort_sa_construction semantics (comparison
 * sort of all suffixes, sentinel smallest); O(n^2 log n), small inputs only. */
int oracle_sa_naive(const uint8_t *text, size_t n, uint32_t *sa_out);

/* stralg/bwt.c:13-20: bwt[i] = SA[i] == 0 ? 0 : text[SA[i] - 1], i in [0, N). */
void oracle_bwt(const uint8_t *text, const uint32_t *sa, size_t N, uint8_t *bwt_out);

/* stralg/bwt.c:35-45: C[a] = number of symbols < a among text[0..N) where
 * text[N-1] is the sentinel 0 (it is counted).  c_out has sigma entries. */
void oracle_c_table(const uint8_t *text, size_t N, uint32_t sigma, uint32_t *c_out);

/* stralg/bwt.c:47-65: O(a,i) = #{k < i : bwt[k] == a}, i in [0, N], stored
 * position-major: o_out[i * sigma + a]; (N+1)*sigma entries, size_t maths
 * (the reference's uint32_t o_size overflows beyond ~204.8 Mi for sigma 5). */
void oracle_o_table(const uint8_t *text, const uint32_t *sa, size_t N, uint32_t sigma,
                    uint32_t *o_out);

/* stralg/remap.c:8-31,102-114: order-preserving dense relabel, 0 reserved.
 * Writes n+1 bytes (incl. the 0 terminator) and the 256-entry table (-1 for
 * absent symbols); returns alphabet_size (distinct symbols + 1), or 0 when
 * more than 127 distinct symbols are present (remap.h:14-18). */
uint32_t oracle_remap(const uint8_t *in, size_t n, uint8_t *out, int16_t table_out[256]);

/* bioinf/fasta.c:26-70,92-135 load_fasta_records' in-place packing, out of place: the image
 * "name\0sequence\0name\0sequence\0..." in file order (the reference's record list is this in
 * reverse).  A header line loses every '>', ' ' and '\t'; a sequence loses all white space and
 * ends at the next '>' wherever it stands.  packed_out needs len + 1 bytes.  Returns 0, or 1 for
 * a file that ends inside a header line (MALFORMED_FILE; what was packed so far is reported). */
int oracle_fasta_pack(const uint8_t *file, size_t len, uint8_t *packed_out, size_t *packed_len, uint32_t *n_records);

/* stralg/suffix_array.c:53-60 compute_inverse: inv[sa[i]] = i. */
void oracle_inverse(const uint32_t *sa, size_t N, uint32_t *inv_out);

/* stralg/suffix_array.c:62-85 compute_lcp (Kasai): lcp[0] = 0, lcp[j] = length of the longest common
 * prefix of the suffixes sa[j-1] and sa[j].  text holds N-1 symbols (the sentinel is implicit). */
void oracle_lcp(const uint8_t *text, const uint32_t *sa, size_t N, uint32_t *lcp_out);

/* stralg/bwt.c:164-199 init_bwt_exact_match_iter: the final (L, R) of the backward search of one
 * remapped pattern of length m; o is position-major with (N+1)*sigma entries. */
void oracle_bwt_exact_search(const uint32_t *c, const uint32_t *o, size_t N, uint32_t sigma,
                             const uint8_t *pattern, size_t m, uint32_t *l_out, uint32_t *r_out);

/* O(n) verifier independent of any construction algorithm: 1 if sa is a
 * permutation of 0..n with strictly increasing suffixes, else 0. */
int oracle_check_sa(const uint8_t *text, size_t n, const uint32_t *sa);

/* Recursion shape of the last oracle_sa_is call on this thread: level sizes
 * n_l and LMS counts m_l (used for the algorithmic-bytes figure of
 * SURVEY.md section 8d).  Returns the number of levels written. */
int oracle_last_levels(uint64_t *n_out, uint64_t *m_out, int cap);

/* splitmix64-driven synthetic inputs shared by tests, bench and the GPU
 * generator (stralg_amd/synth.py restates the same stream in numpy):
 * symbol i = 1 + (splitmix64(seed + i) % (sigma - 1)). */
void oracle_synth(uint8_t *out, size_t n, uint32_t sigma, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
