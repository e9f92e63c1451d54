/*
 * oracle.c -- CPU restatement of stralg's SA-IS / BWT-table path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity status: PINNED against
 * oracle/_ref (the unmodified reference) and tests/golden/.
 *
 * The algorithm is the reference's, pass for pass; the code is a fresh
 * restatement: per-level allocations sized exactly (the reference bumps
 * pointers through 2N-sized slabs, sa_is.c:370-377,484-491), size_t
 * arithmetic throughout, and no leaked names buffer (sa_is.c:485 vs 499-506).
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

#define EMPTY 0xFFFFFFFFu /* sa_is.c:16  UNDEFINED == ~0 */

static __thread uint64_t g_level_n[64];
static __thread uint64_t g_level_m[64];
static __thread int g_levels;

/* ---- per-pass restatements ------------------------------------------- */

/* sa_is.c:134-153 classify_SL: is_s[i] = 1 for S-type, 0 for L-type,
 * right-to-left; the sentinel position n is S, n-1 is L. */
static void pass_types(const uint32_t *x, size_t n, uint8_t *is_s)
{
    is_s[n] = 1;
    for (size_t i = n; i-- > 0;) {
        if (x[i] < x[i + 1])
            is_s[i] = 1;
        else if (x[i] > x[i + 1])
            is_s[i] = 0;
        else
            is_s[i] = is_s[i + 1];
    }
}

/* sa_is.c:155-162 is_LMS_index */
static inline int lms_at(const uint8_t *is_s, size_t i)
{
    return i > 0 && is_s[i] && !is_s[i - 1];
}

/* sa_is.c:164-174 compute_buckets: sizes[c] = #{i <= n : x[i] == c} */
static void pass_bucket_sizes(const uint32_t *x, size_t n, uint32_t sigma, uint32_t *sizes)
{
    memset(sizes, 0, (size_t)sigma * sizeof *sizes);
    for (size_t i = 0; i <= n; ++i)
        sizes[x[i]]++;
}

/* sa_is.c:176-187 find_buckets_beginnings (exclusive prefix sum) */
static void bucket_heads(const uint32_t *sizes, uint32_t sigma, uint32_t *heads)
{
    uint32_t acc = 0;
    for (uint32_t c = 0; c < sigma; ++c) {
        heads[c] = acc;
        acc += sizes[c];
    }
}

/* sa_is.c:189-201 find_buckets_ends (inclusive prefix sum) */
static void bucket_tails(const uint32_t *sizes, uint32_t sigma, uint32_t *tails)
{
    uint32_t acc = 0;
    for (uint32_t c = 0; c < sigma; ++c) {
        acc += sizes[c];
        tails[c] = acc;
    }
}

/* sa_is.c:203-218 place_LMS: text order, each LMS position takes the last
 * free slot of its bucket. */
static void pass_place_lms(const uint32_t *x, size_t n, uint32_t sigma, const uint8_t *is_s,
                           const uint32_t *sizes, uint32_t *cursor, uint32_t *SA)
{
    bucket_tails(sizes, sigma, cursor);
    for (size_t i = 0; i <= n; ++i)
        if (lms_at(is_s, i))
            SA[--cursor[x[i]]] = (uint32_t)i;
}

/* sa_is.c:220-242 induce_L: left-to-right; EMPTY and 0 entries induce nothing */
static void pass_induce_l(const uint32_t *x, size_t n, uint32_t sigma, const uint8_t *is_s,
                          const uint32_t *sizes, uint32_t *cursor, uint32_t *SA)
{
    bucket_heads(sizes, sigma, cursor);
    for (size_t i = 0; i <= n; ++i) {
        uint32_t p = SA[i];
        if (p == EMPTY || p == 0)
            continue;
        uint32_t j = p - 1;
        if (!is_s[j])
            SA[cursor[x[j]]++] = j;
    }
}

/* sa_is.c:245-263 induce_S: right-to-left; no EMPTY test (every slot the
 * cursor reaches has been written by then; EMPTY - 1 would index out of
 * range otherwise, so the guard below only protects the oracle itself). */
static void pass_induce_s(const uint32_t *x, size_t n, uint32_t sigma, const uint8_t *is_s,
                          const uint32_t *sizes, uint32_t *cursor, uint32_t *SA)
{
    bucket_tails(sizes, sigma, cursor);
    for (size_t i = n + 1; i-- > 0;) {
        uint32_t p = SA[i];
        if (p == 0 || p == EMPTY)
            continue;
        uint32_t j = p - 1;
        if (is_s[j])
            SA[--cursor[x[j]]] = j;
    }
}

/* sa_is.c:265-292 equal_LMS: same symbols and same LMS boundaries */
static int same_lms_substring(const uint32_t *x, size_t n, const uint8_t *is_s, size_t a,
                              size_t b)
{
    if (a == n || b == n) /* the sentinel substring is unique */
        return 0;
    for (size_t k = 0;; ++k) {
        int ea = lms_at(is_s, a + k), eb = lms_at(is_s, b + k);
        if (k > 0 && ea && eb)
            return 1;
        if (ea != eb || x[a + k] != x[b + k])
            return 0;
    }
}

/* sa_is.c:295-336 reduce_SA: name LMS substrings in SA order, then compact
 * the names in text order.  Returns the number of LMS positions (incl. the
 * sentinel); *n_names = largest name + 1. */
static size_t pass_name_and_reduce(const uint32_t *x, size_t n, const uint8_t *is_s,
                                   const uint32_t *SA, uint32_t *names, uint32_t *red,
                                   uint32_t *offsets, uint32_t *n_names)
{
    memset(names, 0xFF, (n + 1) * sizeof *names);
    uint32_t name = 0;
    size_t prev = SA[0]; /* == n: the sentinel suffix */
    names[prev] = 0;
    for (size_t i = 1; i <= n; ++i) {
        size_t j = SA[i];
        if (!lms_at(is_s, j))
            continue;
        if (!same_lms_substring(x, n, is_s, prev, j))
            ++name;
        prev = j;
        names[j] = name;
    }
    *n_names = name + 1;
    size_t m = 0;
    for (size_t i = 0; i <= n; ++i) {
        if (names[i] == EMPTY)
            continue;
        offsets[m] = (uint32_t)i;
        red[m] = names[i];
        ++m;
    }
    return m;
}

/* sa_is.c:443-464 remap_LMS: right-to-left over the reduced SA */
static void pass_place_sorted_lms(const uint32_t *x, size_t n, uint32_t sigma,
                                  const uint32_t *sizes, uint32_t *cursor, const uint32_t *SA1,
                                  const uint32_t *offsets, size_t m, uint32_t *SA)
{
    bucket_tails(sizes, sigma, cursor);
    for (size_t i = m; i-- > 0;) {
        uint32_t p = offsets[SA1[i]];
        SA[--cursor[x[p]]] = p;
    }
    SA[0] = (uint32_t)n;
}

/* ---- recursion driver ------------------------------------------------- */

/* sa_is.c:401-441 sort_SA + 340-399 recursive_sorting */
static int sais_level(const uint32_t *x, size_t n, uint32_t sigma, uint32_t *SA, int depth,
                      int allow_shortcut)
{
    if (depth < 64) {
        g_level_n[depth] = n + 1;
        g_level_m[depth] = 0;
        g_levels = depth + 1;
    }
    if (n == 0) { /* sa_is.c:413-417 */
        SA[0] = 0;
        return 0;
    }
    if (allow_shortcut && (size_t)sigma == n + 1) { /* sa_is.c:423-428 */
        SA[0] = (uint32_t)n;
        for (size_t i = 0; i < n; ++i)
            SA[x[i]] = (uint32_t)i;
        return 0;
    }

    int rc = -1;
    uint8_t *is_s = malloc(n + 1);
    uint32_t *sizes = malloc((size_t)sigma * sizeof *sizes);
    uint32_t *cursor = malloc((size_t)sigma * sizeof *cursor);
    uint32_t *names = malloc((n + 1) * sizeof *names);
    uint32_t *red = NULL, *offsets = NULL, *SA1 = NULL;
    if (!is_s || !sizes || !cursor || !names)
        goto out;

    pass_types(x, n, is_s);
    pass_bucket_sizes(x, n, sigma, sizes);

    memset(SA, 0xFF, (n + 1) * sizeof *SA); /* sa_is.c:355 */
    pass_place_lms(x, n, sigma, is_s, sizes, cursor, SA);
    pass_induce_l(x, n, sigma, is_s, sizes, cursor, SA);
    pass_induce_s(x, n, sigma, is_s, sizes, cursor, SA);

    size_t m_cap = n / 2 + 2;
    red = malloc(m_cap * sizeof *red);
    offsets = malloc(m_cap * sizeof *offsets);
    if (!red || !offsets)
        goto out;
    uint32_t n_names = 0;
    size_t m = pass_name_and_reduce(x, n, is_s, SA, names, red, offsets, &n_names);
    if (depth < 64)
        g_level_m[depth] = m;
    free(names);
    names = NULL;

    SA1 = malloc(m * sizeof *SA1);
    if (!SA1)
        goto out;
    /* sa_is.c:335: the reduced string excludes its sentinel from its length */
    if (sais_level(red, m - 1, n_names, SA1, depth + 1, 1) != 0)
        goto out;

    memset(SA, 0xFF, (n + 1) * sizeof *SA); /* sa_is.c:389 */
    pass_place_sorted_lms(x, n, sigma, sizes, cursor, SA1, offsets, m, SA);
    pass_induce_l(x, n, sigma, is_s, sizes, cursor, SA);
    pass_induce_s(x, n, sigma, is_s, sizes, cursor, SA);
    rc = 0;
out:
    free(SA1);
    free(offsets);
    free(red);
    free(names);
    free(cursor);
    free(sizes);
    free(is_s);
    return rc;
}

static int sa_is_entry(const uint8_t *text, size_t n, uint32_t sigma, uint32_t *sa_out,
                       int allow_shortcut)
{
    /* sa_is.c:477-481: widen to u32 with the sentinel appended */
    uint32_t *x = malloc((n + 1) * sizeof *x);
    if (!x)
        return -1;
    for (size_t i = 0; i < n; ++i)
        x[i] = text[i];
    x[n] = 0;
    g_levels = 0;
    int rc = sais_level(x, n, sigma, sa_out, 0, allow_shortcut);
    free(x);
    return rc;
}

int oracle_sa_is(const uint8_t *text, size_t n, uint32_t sigma, uint32_t *sa_out)
{
    return sa_is_entry(text, n, sigma, sa_out, 1);
}

int oracle_sa_is_strict(const uint8_t *text, size_t n, uint32_t sigma, uint32_t *sa_out)
{
    return sa_is_entry(text, n, sigma, sa_out, 0);
}

int oracle_last_levels(uint64_t *n_out, uint64_t *m_out, int cap)
{
    int k = g_levels < cap ? g_levels : cap;
    for (int i = 0; i < k; ++i) {
        n_out[i] = g_level_n[i];
        m_out[i] = g_level_m[i];
    }
    return k;
}

/* ---- naive construction (suffix_array.c:32-48) ------------------------- */

static const uint8_t *g_naive_text;
static size_t g_naive_n;

static int naive_cmp(const void *pa, const void *pb)
{
    size_t a = *(const uint32_t *)pa, b = *(const uint32_t *)pb;
    if (a == b)
        return 0;
    const uint8_t *t = g_naive_text;
    size_t n = g_naive_n;
    while (a < n && b < n) {
        if (t[a] != t[b])
            return t[a] < t[b] ? -1 : 1;
        ++a;
        ++b;
    }
    return a == n ? -1 : 1; /* the one that hits the sentinel first is smaller */
}

int oracle_sa_naive(const uint8_t *text, size_t n, uint32_t *sa_out)
{
    for (size_t i = 0; i <= n; ++i)
        sa_out[i] = (uint32_t)i;
    g_naive_text = text;
    g_naive_n = n;
    qsort(sa_out, n + 1, sizeof *sa_out, naive_cmp);
    return 0;
}

/* ---- BWT tables --------------------------------------------------------- */

void oracle_bwt(const uint8_t *text, const uint32_t *sa, size_t N, uint8_t *bwt_out)
{
    for (size_t i = 0; i < N; ++i)
        bwt_out[i] = sa[i] == 0 ? 0 : text[sa[i] - 1];
}

void oracle_c_table(const uint8_t *text, size_t N, uint32_t sigma, uint32_t *c_out)
{
    uint32_t *cnt = calloc(sigma ? sigma : 1, sizeof *cnt);
    cnt[0] = 1; /* the sentinel at text[N-1] */
    for (size_t i = 0; i + 1 < N; ++i)
        cnt[text[i]]++;
    uint32_t acc = 0;
    for (uint32_t a = 0; a < sigma; ++a) {
        c_out[a] = acc;
        acc += cnt[a];
    }
    free(cnt);
}

void oracle_o_table(const uint8_t *text, const uint32_t *sa, size_t N, uint32_t sigma,
                    uint32_t *o_out)
{
    /* bwt.c:58-65 loops letter-outer; the values are the same row by row */
    for (uint32_t a = 0; a < sigma; ++a)
        o_out[a] = 0;
    for (size_t i = 1; i <= N; ++i) {
        const uint32_t *prev = o_out + (i - 1) * (size_t)sigma;
        uint32_t *row = o_out + i * (size_t)sigma;
        memcpy(row, prev, (size_t)sigma * sizeof *row);
        uint8_t b = sa[i - 1] == 0 ? 0 : text[sa[i - 1] - 1];
        row[b]++;
    }
}

/* ---- remap ---------------------------------------------------------------- */

uint32_t oracle_remap(const uint8_t *in, size_t n, uint8_t *out, int16_t table_out[256])
{
    int seen[256] = {0};
    for (size_t i = 0; i < n; ++i)
        seen[in[i]] = 1;
    uint32_t next = 1;
    table_out[0] = 0;
    for (int c = 1; c < 256; ++c)
        table_out[c] = seen[c] ? (int16_t)next++ : -1;
    if (next > 128)
        return 0;
    for (size_t i = 0; i < n; ++i)
        out[i] = (uint8_t)table_out[in[i]];
    out[n] = 0;
    return next;
}

/* ---- FASTA ingest --------------------------------------------------------------- */

/* isspace() in the C locale, which is what the reference's pack_seq sees */
static int fasta_space(uint8_t c)
{
    return c == ' ' || (c >= '\t' && c <= '\r');
}

int oracle_fasta_pack(const uint8_t *file, size_t len, uint8_t *packed_out, size_t *packed_len, uint32_t *n_records)
{
    /* the reference works on the NUL-terminated image load_file returns (io.c:15-18): a NUL inside the file ends it */
    size_t end = 0;
    while (end < len && file[end] != 0)
        ++end;
#define AT(i) ((i) < end ? file[i] : (uint8_t)0)
    size_t front = 0, pack = 0;
    uint32_t recs = 0;
    int have_front = 1;
    while (have_front) { /* fasta.c:117-134 */
        /* pack_name, fasta.c:26-48 */
        for (;;) {
            while (AT(front) == '>' || AT(front) == ' ' || AT(front) == '\t')
                ++front;
            if (AT(front) == 0 || AT(front) == '\n')
                break;
            packed_out[pack++] = AT(front);
            ++front;
        }
        /* The reference packs in place: when nothing has been dropped yet (a first header line without '>',
         * ' ' or '\t'), the terminator it writes here lands on the very newline it examines next, and the
         * file is taken to end inside the header (fasta.c:41-44 after :39). */
        const int clobbered = pack == front;
        packed_out[pack++] = 0;
        if (AT(front) == 0 || clobbered) { /* the file ends inside a header line: fasta.c:121-124 */
            *packed_len = pack;
            *n_records = recs;
            return 1;
        }
        ++front;
        /* pack_seq, fasta.c:50-70 */
        for (;;) {
            while (AT(front) && fasta_space(AT(front)))
                ++front;
            if (AT(front) == 0 || AT(front) == '>')
                break;
            packed_out[pack++] = AT(front);
            ++front;
        }
        packed_out[pack++] = 0;
        if (AT(front) == 0)
            have_front = 0;
        else
            ++front;
        ++recs;
    }
#undef AT
    *packed_len = pack;
    *n_records = recs;
    return 0;
}

/* ---- extended suffix array, exact search ------------------------------------------ */

void oracle_inverse(const uint32_t *sa, size_t N, uint32_t *inv_out)
{
    for (size_t i = 0; i < N; ++i)
        inv_out[sa[i]] = (uint32_t)i;
}

void oracle_lcp(const uint8_t *text, const uint32_t *sa, size_t N, uint32_t *lcp_out)
{
    uint32_t *inv = malloc(N * sizeof *inv);
    const size_t n = N - 1;
    oracle_inverse(sa, N, inv);
    lcp_out[0] = 0;
    size_t l = 0;
    for (size_t i = 0; i < N; ++i) {
        size_t j = inv[i];
        if (j == 0) /* suffix_array.c:74-75: no predecessor; l is left alone */
            continue;
        size_t k = sa[j - 1];
        /* the reference compares through the 0 terminator, which only the shorter suffix holds */
        while (k + l < n && i + l < n && text[k + l] == text[i + l])
            ++l;
        lcp_out[j] = (uint32_t)l;
        l = l > 0 ? l - 1 : 0;
    }
    free(inv);
}

void oracle_bwt_exact_search(const uint32_t *c, const uint32_t *o, size_t N, uint32_t sigma,
                             const uint8_t *pattern, size_t m, uint32_t *l_out, uint32_t *r_out)
{
    uint32_t L = 0, R = (uint32_t)N;
    if (m > N) { /* bwt.c:178-180 */
        R = 0;
        L = 1;
    }
    for (size_t s = m; s-- > 0 && L < R;) {
        uint8_t a = pattern[s];
        L = c[a] + o[(size_t)L * sigma + a];
        R = c[a] + o[(size_t)R * sigma + a];
    }
    *l_out = L;
    *r_out = R;
}

/* ---- verifier ------------------------------------------------------------- */

int oracle_check_sa(const uint8_t *text, size_t n, const uint32_t *sa)
{
    size_t N = n + 1;
    uint32_t *rank = malloc(N * sizeof *rank);
    if (!rank)
        return 0;
    memset(rank, 0xFF, N * sizeof *rank);
    int ok = 1;
    for (size_t i = 0; i < N && ok; ++i) {
        if (sa[i] > n || rank[sa[i]] != EMPTY)
            ok = 0;
        else
            rank[sa[i]] = (uint32_t)i;
    }
    if (ok && sa[0] != n)
        ok = 0;
    /* suffix a < suffix b  <=>  (text[a], rank[a+1]) < (text[b], rank[b+1]) */
    for (size_t i = 1; i + 1 < N && ok; ++i) {
        size_t a = sa[i], b = sa[i + 1];
        uint8_t ca = text[a], cb = text[b]; /* a, b < n here: only sa[0] == n */
        if (ca > cb)
            ok = 0;
        else if (ca == cb && rank[a + 1] >= rank[b + 1])
            ok = 0;
    }
    free(rank);
    return ok;
}

/* ---- synthetic inputs ------------------------------------------------------ */

static inline uint64_t splitmix64_at(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oracle_synth(uint8_t *out, size_t n, uint32_t sigma, uint64_t seed)
{
    uint32_t span = sigma - 1;
    for (size_t i = 0; i < n; ++i)
        out[i] = (uint8_t)(1 + (uint32_t)((splitmix64_at(seed, i) >> 33) % span));
}
