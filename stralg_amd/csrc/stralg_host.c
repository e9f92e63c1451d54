/*
 * stralg_host.c -- host side of the drop-in: stralg's own entry points for the
 * suffix-array / BWT-table construction path (include/stralg_compat.h), in C,
 * on top of the C-ABI shim (include/stralg_amd.h).  Plain host glue only:
 * allocation with malloc/calloc (callers free() these arrays), the byte remap,
 * string reversal and the o_indices pointer tables stay on the CPU, exactly
 * the parts SURVEY.md section 2 marks as host glue.
 */
#include "stralg_compat.h"
#include "stralg_amd.h"

#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- per-thread device context --------------------------------------------- */

static __thread sx_ctx *tls_ctx = NULL;
static __thread int tls_device = -1;

static void die(const char *what, int rc, const sx_ctx *ctx)
{
    fprintf(stderr, "stralg_amd: %s failed (code %d): %s\n", what, rc, ctx ? sx_last_error(ctx) : "");
    fprintf(stderr, "stralg_amd: the reference entry points cannot report errors and there is no CPU "
                    "fallback; aborting\n");
    abort();
}

/* ---- host-side helpers: a few threads for the byte loops over whole records, large result arrays on huge pages ----
 * A 1 GiB record means 1 GiB of remap, 4 + 20 GiB of results and 8 GiB of row pointers on the host: single-threaded
 * byte loops and page faults cost seconds where the device work costs 30 ms. */

static int host_threads(void)
{
    const char *env = getenv("STRALG_AMD_HOST_THREADS");
    if (env && atoi(env) >= 1) return atoi(env) > 64 ? 64 : atoi(env);
    const long cpus = sysconf(_SC_NPROCESSORS_ONLN);
    return cpus >= 32 ? 16 : (cpus >= 4 ? (int)(cpus / 2) : 1);
}

struct range_job {
    void (*fn)(size_t lo, size_t hi, void *arg);
    void *arg;
    size_t lo, hi;
};

static void *range_worker(void *p)
{
    struct range_job *j = p;
    j->fn(j->lo, j->hi, j->arg);
    return NULL;
}

/* loops shorter than this run on the calling thread ($STRALG_AMD_PARALLEL_MIN: the tests force the threaded path) */
static size_t parallel_min(void)
{
    const char *env = getenv("STRALG_AMD_PARALLEL_MIN");
    return env && atol(env) >= 1 ? (size_t)atol(env) : (size_t)4 << 20;
}

/* slices a loop over [0, total) is cut into: at most one per thread, each at least a quarter of the minimum */
static int slice_count(size_t total)
{
    const size_t pmin = parallel_min(), grain = pmin / 4 ? pmin / 4 : 1;
    int nt = host_threads();
    if (total < pmin || nt <= 1) return 1;
    if ((size_t)nt > total / grain) nt = (int)(total / grain);
    return nt < 1 ? 1 : nt;
}

/* fn(lo, hi, arg) over [0, total) in contiguous slices, one per thread; small totals run inline */
static void parallel_ranges(size_t total, void (*fn)(size_t, size_t, void *), void *arg)
{
    const int nt = slice_count(total);
    if (nt <= 1) {
        fn(0, total, arg);
        return;
    }
    pthread_t th[64];
    struct range_job jobs[64];
    bool threaded[64] = {false};
    const size_t per = (total + (size_t)nt - 1) / (size_t)nt;
    for (int t = 0; t < nt; ++t) {
        const size_t lo = (size_t)t * per, hi = lo + per < total ? lo + per : total;
        if (lo >= hi) break;
        jobs[t] = (struct range_job){fn, arg, lo, hi};
        threaded[t] = pthread_create(&th[t], NULL, range_worker, &jobs[t]) == 0;
        if (!threaded[t]) fn(lo, hi, arg); /* no more threads: do the slice here */
    }
    for (int t = 0; t < nt; ++t)
        if (threaded[t]) pthread_join(th[t], NULL);
}

/* result arrays: malloc-family memory (callers free() it); large ones are aligned to 2 MiB and advised to use
 * transparent huge pages, which turns 6 million page faults per 24 GiB into 12 thousand.
 *
 * Round 4: a per-thread cache of large blocks.  The production loop is build_complete_table -> write -> free -> build
 * (tools/readmappers/bwt_readmapper/bwt_readmapper.c:54-62), and for a 1 GiB record completely_free_bwt_table took 1.9 s --
 * longer than the build: 52 GiB of huge pages unmapped -- only for the next build to map and first-touch as many again,
 * which is what bounds its downloads.  The free_* functions of this library now hand blocks of 64 MiB and more to the
 * calling thread's cache instead of to free(), and big_alloc takes a cached block that is large enough (and not more
 * than twice as large) before it asks malloc.  Cached blocks are ordinary malloc blocks: a caller that free()s an array
 * itself bypasses the cache, nothing else changes.  Bounds: 8 blocks a thread, and $STRALG_AMD_HOST_CACHE_GIB (default
 * 64, never more than half the machine's memory; 0 disables the cache) for ALL threads' caches together (round 5: the
 * farm hands tables to caller threads; a cap per thread let several of them hold all of the machine's memory between
 * them) -- a block that does not fit under the process-wide cap goes to free().  A thread's cache goes with
 * stralg_amd_release() and at thread exit; until then its blocks stay resident (the reference's API has no release call:
 * a long-lived thread keeps up to the cap after its last record -- include/stralg_compat.h says so). */
#include <malloc.h>

#define BLOCK_CACHE_SLOTS 8
struct block_cache {
    void *p[BLOCK_CACHE_SLOTS];
    size_t bytes[BLOCK_CACHE_SLOTS];
    size_t total;
};
static size_t cache_total_all = 0; /* bytes in all threads' caches (atomic) */
static pthread_key_t cache_key;
static pthread_once_t cache_key_once = PTHREAD_ONCE_INIT;
static int cache_key_ok = 0;

static void block_cache_drop(void *arg)
{
    struct block_cache *bc = arg;
    if (!bc) return;
    for (int i = 0; i < BLOCK_CACHE_SLOTS; ++i) free(bc->p[i]);
    __atomic_fetch_sub(&cache_total_all, bc->total, __ATOMIC_RELAXED);
    free(bc);
}

static void cache_key_make(void) { cache_key_ok = pthread_key_create(&cache_key, block_cache_drop) == 0; }

static size_t block_cache_cap(void)
{
    static size_t cap = (size_t)-1;
    if (cap != (size_t)-1) return cap;
    const char *env = getenv("STRALG_AMD_HOST_CACHE_GIB");
    size_t gib = env ? (size_t)atol(env) : 64;
    const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
    if (pages > 0 && psz > 0) {
        const size_t half = (size_t)pages / 2 * (size_t)psz >> 30;
        if (gib > half) gib = half;
    }
    cap = gib << 30;
    const char *bytes_env = getenv("STRALG_AMD_HOST_CACHE_BYTES"); /* (the tests: a cap that a few short records reach) */
    if (bytes_env && atol(bytes_env) >= 0) cap = (size_t)atol(bytes_env);
    return cap;
}

static struct block_cache *block_cache_get(bool create)
{
    pthread_once(&cache_key_once, cache_key_make);
    if (!cache_key_ok) return NULL;
    struct block_cache *bc = pthread_getspecific(cache_key);
    if (!bc && create) {
        bc = calloc(1, sizeof *bc);
        if (bc && pthread_setspecific(cache_key, bc) != 0) {
            free(bc);
            bc = NULL;
        }
    }
    return bc;
}

/* blocks of at least this many bytes are cached (64 MiB; $STRALG_AMD_HOST_CACHE_MIN, in bytes, lets the tests cache small ones) */
static size_t block_cache_min(void)
{
    static size_t v = 0;
    if (!v) {
        const char *env = getenv("STRALG_AMD_HOST_CACHE_MIN");
        v = env && atol(env) > 0 ? (size_t)atol(env) : (size_t)64 << 20;
    }
    return v;
}
#define kBlockCacheMin block_cache_min()

/* free() for the large arrays this library allocated: into the calling thread's cache when there is room */
static void big_free(void *p)
{
    if (!p) return;
    const size_t bytes = malloc_usable_size(p);
    struct block_cache *bc = bytes >= kBlockCacheMin && block_cache_cap() ? block_cache_get(true) : NULL;
    while (bc && bytes <= block_cache_cap()) {
        int empty = -1, smallest = -1;
        for (int i = 0; i < BLOCK_CACHE_SLOTS; ++i) {
            if (!bc->p[i]) empty = i;
            else if (smallest < 0 || bc->bytes[i] < bc->bytes[smallest]) smallest = i;
        }
        if (empty >= 0) { /* reserve the bytes under the process-wide cap first; give them back if they do not fit */
            if (__atomic_add_fetch(&cache_total_all, bytes, __ATOMIC_RELAXED) <= block_cache_cap()) {
                bc->p[empty] = p;
                bc->bytes[empty] = bytes;
                bc->total += bytes;
                return;
            }
            __atomic_fetch_sub(&cache_total_all, bytes, __ATOMIC_RELAXED);
        }
        /* no room: a smaller cached block makes way (a caller that went from short records to long ones: the short
         * records' blocks would otherwise sit in the slots for ever while the long ones' are unmapped every time) */
        if (smallest < 0 || bc->bytes[smallest] >= bytes) break;
        free(bc->p[smallest]);
        bc->total -= bc->bytes[smallest];
        __atomic_fetch_sub(&cache_total_all, bc->bytes[smallest], __ATOMIC_RELAXED);
        bc->p[smallest] = NULL;
        bc->bytes[smallest] = 0;
    }
    free(p);
}

static void block_cache_release(void)
{
    struct block_cache *bc = block_cache_get(false);
    if (!bc) return;
    for (int i = 0; i < BLOCK_CACHE_SLOTS; ++i) {
        free(bc->p[i]);
        bc->p[i] = NULL;
        bc->bytes[i] = 0;
    }
    __atomic_fetch_sub(&cache_total_all, bc->total, __ATOMIC_RELAXED);
    bc->total = 0;
}

/* bytes held by all threads' block caches: what the multi-thread bound's test looks at */
size_t stralg_amd_host_cache_bytes(void) { return __atomic_load_n(&cache_total_all, __ATOMIC_RELAXED); }

static void *big_alloc(size_t bytes)
{
    const size_t huge = (size_t)2 << 20;
    if (bytes < 4 * huge && bytes < kBlockCacheMin) return malloc(bytes ? bytes : 1);
    if (bytes >= kBlockCacheMin) {
        struct block_cache *bc = block_cache_get(false);
        if (bc) { /* the smallest cached block that holds it, unless it is more than twice as large */
            int best = -1;
            for (int i = 0; i < BLOCK_CACHE_SLOTS; ++i)
                if (bc->p[i] && bc->bytes[i] >= bytes && bc->bytes[i] / 2 <= bytes && (best < 0 || bc->bytes[i] < bc->bytes[best])) best = i;
            if (best >= 0) {
                void *p = bc->p[best];
                bc->total -= bc->bytes[best];
                __atomic_fetch_sub(&cache_total_all, bc->bytes[best], __ATOMIC_RELAXED);
                bc->p[best] = NULL;
                bc->bytes[best] = 0;
                return p;
            }
        }
    }
    if (bytes < 4 * huge) return malloc(bytes ? bytes : 1);
    void *p = aligned_alloc(huge, (bytes + huge - 1) & ~(huge - 1));
    if (!p) return malloc(bytes);
    (void)madvise(p, (bytes + huge - 1) & ~(huge - 1), MADV_HUGEPAGE);
    return p;
}

/* A thread that exits without calling stralg_amd_release() -- every caller written against the reference's API: it has
 * no such call -- must not take its device workspace (GiBs of HBM) with it: a pthread key whose destructor destroys the
 * thread's context runs at thread exit.  (The main thread's context goes with the process.) */
static pthread_key_t ctx_key;
static pthread_once_t ctx_key_once = PTHREAD_ONCE_INIT;
static int ctx_key_ok = 0;

static void ctx_key_destroy(void *p)
{
    /* the __thread variables of the exiting thread may be gone already: only the key's value is used */
    if (p) sx_ctx_destroy((sx_ctx *)p);
}

static void ctx_key_make(void) { ctx_key_ok = pthread_key_create(&ctx_key, ctx_key_destroy) == 0; }

static void ctx_key_set(sx_ctx *ctx)
{
    pthread_once(&ctx_key_once, ctx_key_make);
    if (ctx_key_ok) (void)pthread_setspecific(ctx_key, ctx);
}

static sx_ctx *thread_ctx(void)
{
    if (tls_ctx) return tls_ctx;
    if (tls_device < 0) {
        const char *env = getenv("STRALG_AMD_DEVICE");
        tls_device = env ? atoi(env) : 0;
    }
    int rc = sx_ctx_create(tls_device, &tls_ctx);
    if (rc != 0) die("sx_ctx_create", rc, NULL);
    ctx_key_set(tls_ctx);
    return tls_ctx;
}

int stralg_amd_set_device(int device)
{
    if (device < 0 || device >= sx_device_count()) return -1;
    if (tls_ctx && tls_device != device) {
        sx_ctx_destroy(tls_ctx);
        tls_ctx = NULL;
        ctx_key_set(NULL);
    }
    tls_device = device;
    return 0;
}

void stralg_amd_release(void)
{
    if (tls_ctx) sx_ctx_destroy(tls_ctx);
    tls_ctx = NULL;
    ctx_key_set(NULL);
    block_cache_release();
}

/* contexts alive in this process (created minus destroyed): what the thread-exit test looks at */
int stralg_amd_live_contexts(void) { return sx_ctx_live_count(); }

/* ---- suffix arrays (stralg/suffix_array_internal.c:7-19, suffix_array.c:12-24) ---- */

struct suffix_array *allocate_sa_(uint8_t *string)
{
    struct suffix_array *sa = malloc(sizeof *sa);
    size_t len = strlen((const char *)string) + 1;
    sa->string = string;
    sa->length = (uint32_t)len;
    sa->array = big_alloc(len * sizeof *sa->array);
    sa->inverse = NULL;
    sa->lcp = NULL;
    return sa;
}

static struct suffix_array *construct_on_device(uint8_t *string, uint32_t alphabet_size, const char *who)
{
    struct suffix_array *sa = allocate_sa_(string);
    sx_ctx *ctx = thread_ctx();
    int rc = sx_sa_build(ctx, string, (uint64_t)sa->length - 1, alphabet_size, sa->array);
    if (rc != 0) die(who, rc, ctx);
    return sa;
}

struct suffix_array *sa_is_construction(uint8_t *remapped_string, uint32_t alphabet_size)
{
    return construct_on_device(remapped_string, alphabet_size, "sa_is_construction");
}

struct suffix_array *sa_is_mem_construction(uint8_t *remapped_string, uint32_t alphabet_size)
{
    return construct_on_device(remapped_string, alphabet_size, "sa_is_mem_construction");
}

struct suffix_array *skew_sa_construction(uint8_t *string)
{
    /* skew.c:375 works on raw bytes with a fixed alphabet of 256 */
    return construct_on_device(string, 256, "skew_sa_construction");
}

void free_suffix_array(struct suffix_array *sa)
{
    big_free(sa->array);
    big_free(sa->inverse);
    big_free(sa->lcp);
    free(sa);
}

void free_complete_suffix_array(struct suffix_array *sa)
{
    big_free(sa->string);
    free_suffix_array(sa);
}

/* ---- extended suffix array (stralg/suffix_array.c:53-85) ------------------------------- */

void compute_inverse(struct suffix_array *sa)
{
    if (sa->inverse) return;
    sa->inverse = malloc((size_t)sa->length * sizeof *sa->inverse);
    sx_ctx *ctx = thread_ctx();
    int rc = sx_sa_inverse_lcp(ctx, sa->string, sa->array, sa->length, sa->inverse, NULL);
    if (rc != 0) die("compute_inverse", rc, ctx);
}

void compute_lcp(struct suffix_array *sa)
{
    if (sa->lcp) return;
    sa->lcp = malloc((size_t)sa->length * sizeof *sa->lcp);
    uint32_t *inv = sa->inverse ? NULL : malloc((size_t)sa->length * sizeof *inv);
    sx_ctx *ctx = thread_ctx();
    int rc = sx_sa_inverse_lcp(ctx, sa->string, sa->array, sa->length, inv, sa->lcp);
    if (rc != 0) die("compute_lcp", rc, ctx);
    if (inv) sa->inverse = inv; /* compute_lcp leaves the inverse behind, as suffix_array.c:69 does */
}

/* ---- remap (stralg/remap.c:8-114,155-165) ----------------------------------------- */

void init_remap_table(struct remap_table *table, const uint8_t *string)
{
    bool present[256] = {false};
    for (const uint8_t *p = string; *p; ++p) present[*p] = true;
    memset(table->table, -1, sizeof table->table);
    memset(table->rev_table, -1, sizeof table->rev_table);
    table->table[0] = 0; /* the sentinel maps to itself */
    table->rev_table[0] = 0;
    uint32_t next = 1;
    for (int c = 1; c < 256; ++c) {
        if (!present[c]) continue;
        /* more than 127 letters do not fit signed char codes (remap.h:14-18) */
        table->table[c] = (signed char)next;
        if (next < 128) table->rev_table[next] = (signed char)c;
        ++next;
    }
    table->alphabet_size = next;
}

struct remap_table *alloc_remap_table(const uint8_t *string)
{
    struct remap_table *table = malloc(sizeof *table);
    init_remap_table(table, string);
    return table;
}

void dealloc_remap_table(struct remap_table *table) { (void)table; }

void free_remap_table(struct remap_table *table) { free(table); }

uint8_t *remap_between(uint8_t *output, const uint8_t *from, const uint8_t *to, struct remap_table *table)
{
    for (; from != to; ++from, ++output) {
        signed char code = table->table[*from];
        *output = (uint8_t)code;
        if (code < 0) return NULL; /* letter not in the table */
    }
    return output;
}

uint8_t *remap_between0(uint8_t *output, const uint8_t *from, const uint8_t *to, struct remap_table *table)
{
    uint8_t *end = remap_between(output, from, to, table);
    if (!end) return NULL;
    *end = 0;
    return end + 1;
}

uint8_t *remap(uint8_t *output, const uint8_t *input, struct remap_table *table)
{
    /* the terminator is mapped too (0 -> 0), as in remap.c:102-114 */
    return remap_between(output, input, input + strlen((const char *)input) + 1, table);
}

uint32_t remap_string(uint8_t *output, uint8_t *input)
{
    struct remap_table table;
    init_remap_table(&table, input);
    remap(output, input, &table);
    return table.alphabet_size;
}

/* ---- BWT tables (stralg/bwt.c:22-161) ------------------------------------------------ */

struct row_fill {
    uint32_t **idx;
    uint32_t *table;
    size_t sigma;
};

static void fill_rows(size_t lo, size_t hi, void *arg)
{
    const struct row_fill *r = arg;
    for (size_t i = lo; i < hi; ++i) r->idx[i] = r->table + r->sigma * i;
}

/* o_indices / ro_indices (bwt.c:52-57): 8 bytes per row, filled by a few threads */
static uint32_t **row_pointers(uint32_t *table, size_t rows, uint32_t sigma)
{
    uint32_t **idx = big_alloc(rows * sizeof *idx);
    struct row_fill r = {idx, table, sigma};
    parallel_ranges(rows, fill_rows, &r);
    return idx;
}

void init_bwt_table(struct bwt_table *bwt_table, struct suffix_array *sa, struct suffix_array *rsa,
                    struct remap_table *remap_table)
{
    const uint32_t sigma = remap_table->alphabet_size;
    const size_t N = sa->length;
    /* size_t, unlike bwt.c:50-51 whose uint32_t o_size wraps beyond ~204.8 Mi symbols */
    const size_t o_words = (size_t)sigma * (N + 1);
    sx_ctx *ctx = thread_ctx();

    bwt_table->remap_table = remap_table;
    bwt_table->sa = sa;
    bwt_table->c_table = calloc(sigma, sizeof *bwt_table->c_table);
    bwt_table->o_table = big_alloc(o_words * sizeof *bwt_table->o_table);
    int rc = sx_bwt_tables(ctx, sa->string, sa->array, N, sigma, bwt_table->c_table, bwt_table->o_table);
    if (rc != 0) die("init_bwt_table", rc, ctx);
    bwt_table->o_indices = row_pointers(bwt_table->o_table, N + 1, sigma);

    if (rsa) {
        uint32_t *c_tmp = calloc(sigma, sizeof *c_tmp);
        bwt_table->ro_table = big_alloc(o_words * sizeof *bwt_table->ro_table);
        rc = sx_bwt_tables(ctx, rsa->string, rsa->array, rsa->length, sigma, c_tmp, bwt_table->ro_table);
        if (rc != 0) die("init_bwt_table (reverse)", rc, ctx);
        free(c_tmp);
        bwt_table->ro_indices = row_pointers(bwt_table->ro_table, N + 1, sigma);
    } else {
        bwt_table->ro_table = NULL;
        bwt_table->ro_indices = NULL;
    }
}

struct bwt_table *alloc_bwt_table(struct suffix_array *sa, struct suffix_array *rsa,
                                  struct remap_table *remap_table)
{
    struct bwt_table *table = malloc(sizeof *table);
    init_bwt_table(table, sa, rsa, remap_table);
    return table;
}

void dealloc_bwt_table(struct bwt_table *bwt_table)
{
    free(bwt_table->c_table);
    big_free(bwt_table->o_table);
    big_free(bwt_table->o_indices);
    big_free(bwt_table->ro_table);
    big_free(bwt_table->ro_indices);
}

void free_bwt_table(struct bwt_table *bwt_table)
{
    dealloc_bwt_table(bwt_table);
    free(bwt_table);
}

void completely_dealloc_bwt_table(struct bwt_table *bwt_table)
{
    free_complete_suffix_array(bwt_table->sa);
    free_remap_table(bwt_table->remap_table);
    dealloc_bwt_table(bwt_table);
}

void completely_free_bwt_table(struct bwt_table *bwt_table)
{
    completely_dealloc_bwt_table(bwt_table);
    free(bwt_table);
}

struct presence_job {
    const uint8_t *string;
    bool present[64][256]; /* one row per slice */
    size_t per;
};

static void presence_slice(size_t lo, size_t hi, void *arg)
{
    struct presence_job *j = arg;
    bool *present = j->present[lo / j->per];
    for (size_t i = lo; i < hi; ++i) present[j->string[i]] = true;
}

struct lut_job {
    const uint8_t *in;
    uint8_t *out;
    const signed char *table;
    size_t n; /* reverse: out[i] = table[in[n - 1 - i]] */
    bool reverse;
};

static void lut_slice(size_t lo, size_t hi, void *arg)
{
    const struct lut_job *j = arg;
    if (j->reverse)
        for (size_t i = lo; i < hi; ++i) j->out[i] = j->in[j->n - 1 - i];
    else
        for (size_t i = lo; i < hi; ++i) j->out[i] = (uint8_t)j->table[j->in[i]];
}

/* ---- the length and the letters of a long record in one pass ---------------------------------------------------------
 * build_complete_table gets a NUL-terminated string: the reference starts with strlen (bwt.c:139) and walks the record
 * again for its letters (remap.c:8-31, 73-77).  Here one walk finds both, on the calling thread, and -- like strlen --
 * never touches a byte behind the terminator's 32-byte block: loads are 32 bytes at 32-byte-aligned addresses, so a load
 * that holds a byte of the string lies in that byte's page.  (Round 4 scanned 4 MiB chunks of the string's mapping on
 * several threads, ahead of the terminator: memory behind a string belongs to whoever allocated it, and another thread
 * unmapping or protecting it between the look at /proc/self/maps and the read was a SIGSEGV this library caused.  The
 * length of a NUL-terminated string cannot be found in parallel without reading ahead of the terminator; callers that know
 * the length -- the batch entry points -- hand it over and get the letters by a parallel pass over [0, n) instead.) */
#define SCAN_CHUNK ((size_t)4 << 20)

/* A record has few distinct letters: 32 bytes at a time are compared with the letters seen so far (up to 16 of them: a
 * compare and an OR each), and only a block that holds a new one is walked byte by byte -- a table look-up per byte ran at
 * a byte a cycle; this form is bound by the core's memory stream. */
#if defined(__x86_64__)
#include <immintrin.h>
/* returns the offset of the first NUL in p[0 .. len) (len: none; len = SIZE_MAX: the string is NUL-terminated), bits |= the
 * byte values in front of it.  Reads whole aligned 32-byte blocks that hold at least one byte of p[0 .. min(len, NUL)]. */
__attribute__((target("avx2"))) static size_t scan_block_avx2(const uint8_t *p, size_t len, uint32_t bits[8])
{
    uint8_t known[16];
    int nk = 0;
    bool many = false; /* more than 16 distinct letters: every block is walked byte by byte */
    for (int c = 1; c < 256 && nk < 16; ++c)
        if ((bits[c >> 5] >> (c & 31)) & 1u) known[nk++] = (uint8_t)c;
    const __m256i zero = _mm256_setzero_si256();
    size_t i = 0;
    /* bytes up to the first aligned address, one at a time */
    for (; i < len && ((uintptr_t)(p + i) & 31u) != 0; ++i) {
        if (!p[i]) return i;
        bits[p[i] >> 5] |= 1u << (p[i] & 31);
    }
    for (; len - i >= 32; i += 32) {
        const __m256i v = _mm256_load_si256((const __m256i *)(p + i));
        const uint32_t z = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, zero));
        if (!z && !many) {
            __m256i m = zero;
            for (int q = 0; q < nk; ++q) m = _mm256_or_si256(m, _mm256_cmpeq_epi8(v, _mm256_set1_epi8((char)known[q])));
            if ((uint32_t)_mm256_movemask_epi8(m) == 0xFFFFFFFFu) continue;
            nk = 0; /* the letters seen so far, again from the bit set (a new one joins below) */
        }
        const int stop = z ? __builtin_ctz(z) : 32;
        for (int e = 0; e < stop; ++e) {
            const uint8_t b = p[i + e];
            bits[b >> 5] |= 1u << (b & 31);
        }
        if (z) return i + (size_t)stop;
        if (!many) {
            for (int c = 1; c < 256 && nk <= 16; ++c)
                if ((bits[c >> 5] >> (c & 31)) & 1u) {
                    if (nk < 16) known[nk] = (uint8_t)c;
                    ++nk;
                }
            if (nk > 16) many = true, nk = 16;
        }
    }
    for (; i < len; ++i) {
        if (!p[i]) return i;
        bits[p[i] >> 5] |= 1u << (p[i] & 31);
    }
    return len;
}
#endif

static size_t scan_block(const uint8_t *p, size_t len, uint32_t bits[8])
{
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) return scan_block_avx2(p, len, bits);
#endif
    const size_t upto = len == (size_t)-1 ? strlen((const char *)p) : strnlen((const char *)p, len);
    for (size_t i = 0; i < upto; ++i) bits[p[i] >> 5] |= 1u << (p[i] & 31);
    return upto;
}

/* The walk: strnlen over 4 MiB chunks on the calling thread (libc's: the speed of one core's memory stream, 75 ms a GiB
 * where the fused walk above takes 185), and up to three helper threads that collect the letters of chunks the walk has
 * already found free of terminators -- behind the walk, never ahead of it. */
struct follow_job {
    const uint8_t *s;
    size_t clear;  /* bytes from s on known to hold no terminator: a multiple of the chunk (atomic) */
    int done;      /* the walk has found the terminator (atomic) */
    size_t next;   /* ticket: the next chunk a helper takes (atomic) */
    uint32_t bits[4][8];
    int ids;       /* helper numbers (atomic) */
};

static void *follow_worker(void *arg)
{
    struct follow_job *j = arg;
    uint32_t *bits = j->bits[__atomic_fetch_add(&j->ids, 1, __ATOMIC_RELAXED)];
    for (;;) {
        const size_t k = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        for (;;) {
            if (__atomic_load_n(&j->clear, __ATOMIC_ACQUIRE) >= (k + 1) * SCAN_CHUNK) break;
            if (__atomic_load_n(&j->done, __ATOMIC_ACQUIRE)) { /* (clear is final once done is set) */
                if (__atomic_load_n(&j->clear, __ATOMIC_ACQUIRE) >= (k + 1) * SCAN_CHUNK) break;
                return NULL; /* chunk k holds the terminator or lies behind it: the caller takes what is left */
            }
            usleep(40);
        }
        (void)scan_block(j->s + k * SCAN_CHUNK, SCAN_CHUNK, bits);
    }
}

/* strlen(string), and which byte values the string holds (present[256]; may be NULL); *have_letters says whether
 * `present` was filled (short strings leave that to the caller) */
static size_t long_strlen(const uint8_t *string, bool *present, bool *have_letters)
{
    *have_letters = false;
    const size_t head = strnlen((const char *)string, SCAN_CHUNK);
    if (head < SCAN_CHUNK) return head;
    struct follow_job j;
    memset(&j, 0, sizeof j);
    j.s = string;
    j.clear = SCAN_CHUNK;
    pthread_t th[3];
    int started = 0;
    const int helpers = present ? (host_threads() - 1 < 3 ? host_threads() - 1 : 3) : 0;
    for (int t = 0; t < helpers; ++t)
        if (pthread_create(&th[started], NULL, follow_worker, &j) == 0) ++started;
    size_t n = SCAN_CHUNK;
    for (;;) {
        const size_t e = strnlen((const char *)string + n, SCAN_CHUNK);
        n += e;
        if (e < SCAN_CHUNK) break;
        __atomic_store_n(&j.clear, n, __ATOMIC_RELEASE);
    }
    __atomic_store_n(&j.done, 1, __ATOMIC_RELEASE);
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    if (present) {
        /* what no helper took: from the first ticket nobody finished (all of it when no helper started) to the end */
        const size_t taken = started ? __atomic_load_n(&j.next, __ATOMIC_RELAXED) - (size_t)started : 0;
        const size_t from = taken * SCAN_CHUNK < n ? taken * SCAN_CHUNK : n - n % SCAN_CHUNK;
        uint32_t bits[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)scan_block(string + from, n - from, bits);
        for (int t = 0; t < 4; ++t)
            for (int w = 0; w < 8; ++w) bits[w] |= j.bits[t][w];
        for (int c = 0; c < 256; ++c) present[c] = (bits[c >> 5] >> (c & 31)) & 1u;
        *have_letters = true;
    }
    return n;
}

/* long_strlen for the tests: the length, and present[c] = 1 for every byte value c the string holds (when the parallel scan
 * supplied them: the return value's companion *have_letters) */
size_t stralg_amd_strlen_and_letters(const uint8_t *string, uint8_t *present /* 256 */, int *have_letters)
{
    bool letters[256], have = false;
    memset(letters, 0, sizeof letters);
    const size_t n = long_strlen(string, letters, &have);
    if (present)
        for (int c = 0; c < 256; ++c) present[c] = have && letters[c];
    if (have_letters) *have_letters = have;
    return n;
}

/* alloc_remap_table + remap (remap.c:8-41,102-114) for a record of n letters, a few threads on the byte loops */
static struct remap_table *remap_record(const uint8_t *string, size_t n, uint8_t *remapped, const bool *letters /* or NULL */)
{
    struct remap_table *table = malloc(sizeof *table);
    const int slices = slice_count(n);
    if (slices <= 1) {
        init_remap_table(table, string);
        remap(remapped, string, table);
        return table;
    }
    struct presence_job *pj = calloc(1, sizeof *pj);
    if (letters) { /* (long_strlen has looked at every byte already) */
        memcpy(pj->present[0], letters, 256);
    } else {
        pj->string = string;
        pj->per = (n + (size_t)slices - 1) / (size_t)slices; /* the slicing parallel_ranges uses */
        parallel_ranges(n, presence_slice, pj);
    }
    memset(table->table, -1, sizeof table->table);
    memset(table->rev_table, -1, sizeof table->rev_table);
    table->table[0] = 0;
    table->rev_table[0] = 0;
    uint32_t next = 1;
    for (int c = 1; c < 256; ++c) {
        bool seen = false;
        for (int t = 0; t < 64; ++t) seen = seen || pj->present[t][c];
        if (!seen) continue;
        table->table[c] = (signed char)next;
        if (next < 128) table->rev_table[next] = (signed char)c;
        ++next;
    }
    table->alphabet_size = next;
    free(pj);
    if (next <= 128) {
        struct lut_job lj = {string, remapped, table->table, n, false};
        parallel_ranges(n, lut_slice, &lj);
        remapped[n] = 0;
    }
    return table;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

/* (Round 3 measured two ways of hiding host work behind the downloads of a 1 GiB record, both on the GPU box: the
 * row-pointer tables filled by helper threads while sx_build_tables runs -- the download slowed down by what the fill
 * took, 660 against 515 ms, the sum unchanged at 765 - 800 ms -- and the whole reverse direction through a second
 * context beside the forward one -- 1480 - 1530 ms either way.  The host side is bound by first-touch page faults and
 * the memory bandwidth of the GPU's NUMA node (zeroing + DMA + pointer fill: 64 GiB of writes a record), not by idle
 * time, so the phases stay one after the other.  $STRALG_AMD_TIMING=1 prints them.) */
/* build_complete_table with an error channel: 0 and *out, or a code (SX_E_ARG: more letters than a remap table holds;
 * SX_E_NOMEM; a device error) with everything released and a line on stderr.  The reference-named entry point below
 * aborts on a code (the reference's constructors cannot fail); the farm keeps going with the
 * other records. */
static void free_partial_table(struct bwt_table *table)
{
    if (!table) return;
    if (table->sa) {
        free(table->sa->string);
        free(table->sa->array);
        free(table->sa);
    }
    free(table->remap_table);
    free(table->c_table);
    free(table->o_table);
    free(table->o_indices);
    free(table->ro_table);
    free(table->ro_indices);
    free(table);
}

#define LENGTH_UNKNOWN ((size_t)-1)
/* known_n: strlen(string) where the caller has it (the batch entry points: the letters then come from a parallel pass over
 * [0, n)), LENGTH_UNKNOWN otherwise (one walk on this thread finds the terminator and the letters) */
static int build_complete_table_try(const uint8_t *string, size_t known_n, bool include_reverse, struct bwt_table **out)
{
    *out = NULL;
    const bool timing = getenv("STRALG_AMD_TIMING") != NULL;
    const double t0 = timing ? now_ms() : 0.0;
    bool letters[256], have_letters = false;
    const size_t n = known_n != LENGTH_UNKNOWN ? known_n : long_strlen(string, letters, &have_letters);
    uint8_t *remapped = big_alloc(n + 1);
    if (!remapped) return SX_E_NOMEM;
    struct remap_table *remap_table = remap_record(string, n, remapped, have_letters ? letters : NULL);
    if (remap_table->alphabet_size > 128) {
        fprintf(stderr, "stralg_amd: build_complete_table: %u distinct letters; stralg's remap table holds "
                        "at most 127 (stralg/remap.h:14-18)\n", remap_table->alphabet_size - 1);
        free(remapped);
        free(remap_table);
        return SX_E_ARG;
    }
    const uint32_t sigma = remap_table->alphabet_size;
    const size_t N = n + 1;
    const size_t o_words = (size_t)sigma * (N + 1);
    sx_ctx *ctx = thread_ctx();

    /* One device pass per direction (sx_build_tables): the induced sort hands the BWT over with
     * the suffix array, so init_bwt_table's gather of text[SA[i]-1] is not repeated.  The
     * results are what sa_is_construction + init_bwt_table produce (bwt.c:143-154). */
    struct suffix_array *sa = calloc(1, sizeof *sa); /* allocate_sa_ without its strlen; `remapped` moves into sa->string */
    struct bwt_table *table = calloc(1, sizeof *table);
    if (!sa || !table) {
        free(sa), free(table), free(remapped), free(remap_table);
        return SX_E_NOMEM;
    }
    sa->string = remapped;
    sa->length = (uint32_t)N;
    sa->array = big_alloc(N * sizeof *sa->array);
    table->remap_table = remap_table;
    table->sa = sa;
    table->c_table = calloc(sigma, sizeof *table->c_table);
    table->o_table = big_alloc(o_words * sizeof *table->o_table);
    if (!sa->array || !table->c_table || !table->o_table) {
        fprintf(stderr, "stralg_amd: build_complete_table: no host memory for the tables of %zu symbols\n", n);
        free_partial_table(table);
        return SX_E_NOMEM;
    }
    const double t1 = timing ? now_ms() : 0.0;
    int rc = sx_build_tables(ctx, remapped, n, sigma, sa->array, table->c_table, table->o_table);
    if (rc != 0) {
        fprintf(stderr, "stralg_amd: build_complete_table failed (code %d): %s\n", rc, sx_last_error(ctx));
        free_partial_table(table);
        return rc;
    }
    const double t2 = timing ? now_ms() : 0.0;
    table->o_indices = row_pointers(table->o_table, N + 1, sigma);
    if (timing)
        fprintf(stderr, "stralg_amd timing: n=%zu strlen + remap + allocation %.1f ms, sx_build_tables %.1f ms, row pointers "
                        "%.1f ms\n", n, t1 - t0, t2 - t1, now_ms() - t2);

    table->ro_table = NULL;
    table->ro_indices = NULL;
    if (include_reverse) {
        /* the reverse suffix array and the reversed copy are temporary (bwt.c:147-158) */
        const double t3 = timing ? now_ms() : 0.0;
        uint8_t *rev = big_alloc(n + 1);
        uint32_t *c_tmp = calloc(sigma, sizeof *c_tmp);
        table->ro_table = big_alloc(o_words * sizeof *table->ro_table);
        if (!rev || !c_tmp || !table->ro_table) {
            fprintf(stderr, "stralg_amd: build_complete_table: no host memory for the reverse table of %zu symbols\n", n);
            free(rev), free(c_tmp);
            free_partial_table(table);
            return SX_E_NOMEM;
        }
        struct lut_job lj = {remapped, rev, NULL, n, true};
        parallel_ranges(n, lut_slice, &lj);
        rev[n] = 0;
        rc = sx_build_tables(ctx, rev, n, sigma, NULL, c_tmp, table->ro_table);
        free(c_tmp);
        big_free(rev);
        if (rc != 0) {
            fprintf(stderr, "stralg_amd: build_complete_table (reverse) failed (code %d): %s\n", rc, sx_last_error(ctx));
            free_partial_table(table);
            return rc;
        }
        table->ro_indices = row_pointers(table->ro_table, N + 1, sigma);
        if (timing) fprintf(stderr, "stralg_amd timing: reverse direction %.1f ms\n", now_ms() - t3);
    }
    *out = table;
    return 0;
}

struct bwt_table *build_complete_table(const uint8_t *string, bool include_reverse)
{
    struct bwt_table *table = NULL;
    const int rc = build_complete_table_try(string, LENGTH_UNKNOWN, include_reverse, &table);
    if (rc != 0) die("build_complete_table", rc, tls_ctx);
    return table;
}

/* ---- exact search over host tables (stralg/bwt.c:164-223) -------------------------------- */

void init_bwt_exact_match_iter(struct bwt_exact_match_iter *iter, struct bwt_table *bwt_table,
                               const uint8_t *remapped_pattern)
{
    const struct suffix_array *sa = bwt_table->sa;
    const size_t m = strlen((const char *)remapped_pattern);
    uint32_t L = 0, R = sa->length;
    if (m > sa->length) { /* bwt.c:178-180 */
        R = 0;
        L = 1;
    }
    for (size_t s = m; s-- > 0 && L < R;) {
        const uint8_t a = remapped_pattern[s];
        L = bwt_table->c_table[a] + bwt_table->o_indices[L][a];
        R = bwt_table->c_table[a] + bwt_table->o_indices[R][a];
    }
    iter->sa = sa;
    iter->L = L;
    iter->R = R;
    iter->i = L;
}

bool next_bwt_exact_match_iter(struct bwt_exact_match_iter *iter, struct bwt_exact_match *match)
{
    if (iter->i < 0 || iter->i >= iter->R) return false;
    match->pos = iter->sa->array[iter->i];
    iter->i++;
    return true;
}

void dealloc_bwt_exact_match_iter(struct bwt_exact_match_iter *iter) { (void)iter; }

/* ---- index serialisation (stralg/serialise.c:7-39, string_utils.c:48-96, suffix_array.c:238-258,
 *      remap.c:168-187, bwt.c:425-487).  The byte format is the reference's; lengths that it
 *      computes in 32 bits (o_table_length) are computed in size_t. ------------------------------ */

void write_string_len(FILE *f, const uint8_t *str, uint32_t len)
{
    fwrite(&len, sizeof len, 1, f);
    fwrite(str, 1, len, f);
}

void write_string(FILE *f, const uint8_t *str) { write_string_len(f, str, (uint32_t)strlen((const char *)str) + 1); }

/* the _fname siblings (string_utils.c:54-71,84-96, suffix_array.c:243-267, remap.c:175-201, bwt.c:443-503):
 * open, call the FILE form, close; like the reference they do not check fopen */
void write_string_len_fname(const char *fname, const uint8_t *str, uint32_t len)
{
    FILE *f = fopen(fname, "wb");
    write_string_len(f, str, len);
    fclose(f);
}

void write_string_fname(const char *fname, const uint8_t *str)
{
    write_string_len_fname(fname, str, (uint32_t)strlen((const char *)str) + 1);
}

uint8_t *read_string_len(FILE *f, uint32_t *len)
{
    uint32_t str_len = 0;
    if (fread(&str_len, sizeof str_len, 1, f) != 1) str_len = 0;
    *len = str_len;
    uint8_t *str = malloc((size_t)str_len + 1);
    if (str_len && fread(str, 1, str_len, f) != str_len) memset(str, 0, str_len);
    str[str_len] = 0;
    return str;
}

uint8_t *read_string(FILE *f)
{
    uint32_t dummy;
    return read_string_len(f, &dummy);
}

uint8_t *read_string_len_fname(const char *fname, uint32_t *len)
{
    FILE *f = fopen(fname, "rb");
    uint8_t *str = read_string_len(f, len);
    fclose(f);
    return str;
}

uint8_t *read_string_fname(const char *fname)
{
    uint32_t dummy;
    return read_string_len_fname(fname, &dummy);
}

void write_suffix_array(FILE *f, const struct suffix_array *sa) { fwrite(sa->array, sizeof *sa->array, sa->length, f); }

struct suffix_array *read_suffix_array(FILE *f, uint8_t *string)
{
    struct suffix_array *sa = allocate_sa_(string);
    if (fread(sa->array, sizeof *sa->array, sa->length, f) != sa->length) memset(sa->array, 0, sizeof *sa->array * sa->length);
    return sa;
}

void write_suffix_array_fname(const char *fname, const struct suffix_array *sa)
{
    FILE *f = fopen(fname, "wb");
    write_suffix_array(f, sa);
    fclose(f);
}

struct suffix_array *read_suffix_array_fname(const char *fname, uint8_t *string)
{
    FILE *f = fopen(fname, "rb");
    struct suffix_array *sa = read_suffix_array(f, string);
    fclose(f);
    return sa;
}

void write_remap_table(FILE *f, const struct remap_table *table) { fwrite(table, sizeof *table, 1, f); }

void write_remap_table_fname(const char *fname, const struct remap_table *table)
{
    FILE *f = fopen(fname, "wb");
    write_remap_table(f, table);
    fclose(f);
}

struct remap_table *read_remap_table(FILE *f)
{
    struct remap_table *table = malloc(sizeof *table);
    if (fread(table, sizeof *table, 1, f) != 1) memset(table, 0, sizeof *table);
    return table;
}

struct remap_table *read_remap_table_fname(const char *fname)
{
    FILE *f = fopen(fname, "rb");
    struct remap_table *table = read_remap_table(f);
    fclose(f);
    return table;
}

void write_bwt_table(FILE *f, const struct bwt_table *bwt_table)
{
    const size_t sigma = bwt_table->remap_table->alphabet_size;
    const size_t o_words = sigma * ((size_t)bwt_table->sa->length + 1);
    fwrite(bwt_table->c_table, sizeof *bwt_table->c_table, sigma, f);
    fwrite(bwt_table->o_table, sizeof *bwt_table->o_table, o_words, f);
    const bool has_ro_table = bwt_table->ro_table != NULL;
    fwrite(&has_ro_table, sizeof has_ro_table, 1, f);
    if (has_ro_table) fwrite(bwt_table->ro_table, sizeof *bwt_table->ro_table, o_words, f);
}

struct bwt_table *read_bwt_table(FILE *f, struct suffix_array *sa, struct remap_table *remap_table)
{
    struct bwt_table *table = malloc(sizeof *table);
    table->remap_table = remap_table;
    table->sa = sa;
    const size_t sigma = remap_table->alphabet_size, rows = (size_t)sa->length + 1, o_words = sigma * rows;
    table->c_table = malloc(sigma * sizeof *table->c_table);
    table->o_table = malloc(o_words * sizeof *table->o_table);
    bool ok = fread(table->c_table, sizeof *table->c_table, sigma, f) == sigma;
    ok = ok && fread(table->o_table, sizeof *table->o_table, o_words, f) == o_words;
    table->o_indices = row_pointers(table->o_table, rows, (uint32_t)sigma);
    table->ro_table = NULL;
    table->ro_indices = NULL;
    bool has_ro_table = false;
    if (!ok || fread(&has_ro_table, sizeof has_ro_table, 1, f) != 1) has_ro_table = false;
    if (has_ro_table) {
        table->ro_table = malloc(o_words * sizeof *table->ro_table);
        if (fread(table->ro_table, sizeof *table->ro_table, o_words, f) != o_words) memset(table->ro_table, 0, o_words * 4);
        table->ro_indices = row_pointers(table->ro_table, rows, (uint32_t)sigma);
    }
    return table;
}

void write_bwt_table_fname(const char *fname, const struct bwt_table *bwt_table)
{
    FILE *f = fopen(fname, "wb");
    write_bwt_table(f, bwt_table);
    fclose(f);
}

struct bwt_table *read_bwt_table_fname(const char *fname, struct suffix_array *sa, struct remap_table *remap_table)
{
    FILE *f = fopen(fname, "rb");
    struct bwt_table *table = read_bwt_table(f, sa, remap_table);
    fclose(f);
    return table;
}

void write_complete_bwt_info(FILE *f, const struct bwt_table *bwt_table)
{
    const struct suffix_array *sa = bwt_table->sa;
    write_string_len(f, sa->string, sa->length - 1);
    write_suffix_array(f, sa);
    write_remap_table(f, bwt_table->remap_table);
    write_bwt_table(f, bwt_table);
}

void write_complete_bwt_info_fname(const char *fname, const struct bwt_table *bwt_table)
{
    FILE *f = fopen(fname, "wb");
    write_complete_bwt_info(f, bwt_table);
    fclose(f);
}

struct bwt_table *read_complete_bwt_info(FILE *f)
{
    uint32_t str_len;
    uint8_t *str = read_string_len(f, &str_len);
    struct suffix_array *sa = read_suffix_array(f, str);
    struct remap_table *remap_table = read_remap_table(f);
    return read_bwt_table(f, sa, remap_table);
}

struct bwt_table *read_complete_bwt_info_fname(const char *fname)
{
    FILE *f = fopen(fname, "rb");
    struct bwt_table *res = read_complete_bwt_info(f);
    fclose(f);
    return res;
}

/* build_complete_table + write_complete_bwt_info without the tables ever existing on the host: the suffix
 * array and the O tables stream from the GPU into the file in 32 MiB chunks (sx_build_tables_stream). */
struct stream_sink {
    FILE *f;
    const struct remap_table *remap_table; /* written between the suffix array and the C table; NULL: reverse pass */
    bool remap_written, failed;
};

static int stream_to_file(void *user, int section, const void *data, size_t bytes)
{
    struct stream_sink *s = user;
    if (!s->remap_table && section != SX_SECTION_O) return 0; /* the reverse pass contributes its O table only */
    if (s->remap_table && section != SX_SECTION_SA && !s->remap_written) {
        write_remap_table(s->f, s->remap_table);
        s->remap_written = true;
    }
    if (fwrite(data, 1, bytes, s->f) != bytes) s->failed = true;
    return s->failed ? -1 : 0;
}

int stralg_amd_write_complete_bwt_info_stream(FILE *f, const uint8_t *string, bool include_reverse)
{
    bool letters[256], have_letters = false;
    const size_t n = long_strlen(string, letters, &have_letters);
    uint8_t *remapped = malloc(n + 1);
    struct remap_table *remap_table = remap_record(string, n, remapped, have_letters ? letters : NULL); /* (a few threads for a long record) */
    if (remap_table->alphabet_size > 128) {
        free(remapped);
        free_remap_table(remap_table);
        return -1;
    }
    write_string_len(f, remapped, (uint32_t)n);
    sx_ctx *ctx = thread_ctx();
    struct stream_sink sink = {f, remap_table, false, false};
    int rc = sx_build_tables_stream(ctx, remapped, n, remap_table->alphabet_size, 1, stream_to_file, &sink);
    if (rc == 0) {
        const bool has_ro_table = include_reverse;
        fwrite(&has_ro_table, sizeof has_ro_table, 1, f);
        if (include_reverse) {
            uint8_t *rev = malloc(n + 1);
            for (size_t i = 0; i < n; ++i) rev[i] = remapped[n - 1 - i];
            rev[n] = 0;
            struct stream_sink rsink = {f, NULL, false, false};
            rc = sx_build_tables_stream(ctx, rev, n, remap_table->alphabet_size, 0, stream_to_file, &rsink);
            free(rev);
        }
    }
    free(remapped);
    free_remap_table(remap_table);
    return rc;
}

/* ---- FASTA records (bioinf/fasta.c:92-222) ------------------------------------------------ */

struct fasta_record_impl {
    const char *name;
    const uint8_t *seq;
    uint32_t seq_len;
    uint32_t no_records;
    struct fasta_record_impl *next;
};
struct fasta_records {
    uint8_t *buffer; /* the packed image: name\0sequence\0... in file order */
    struct fasta_record_impl *recs, *storage;
};

struct fasta_records *load_fasta_records(const char *fname, enum error_codes *err)
{
    if (err) *err = NO_ERROR;
    FILE *f = fopen(fname, "rb");
    if (!f) {
        if (err) *err = CANNOT_OPEN_FILE;
        return NULL;
    }
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *image = malloc((size_t)fsize + 1);
    uint8_t *packed = malloc((size_t)fsize + 2);
    if (!image || !packed) die("load_fasta_records: out of memory", -2, NULL);
    if (fsize && fread(image, (size_t)fsize, 1, f) != 1) die("load_fasta_records: read failed", -1, NULL);
    fclose(f);
    /* every record but the first starts at a '>', and every record has two terminators: a bound for the table */
    size_t starts = 1;
    for (const uint8_t *p = image, *end = image + fsize; (p = memchr(p, '>', (size_t)(end - p))) != NULL; ++p) ++starts;
    const uint64_t term_cap = 2 * (uint64_t)starts + 2;
    uint32_t *term = malloc(term_cap * sizeof *term);
    if (!term) die("load_fasta_records: out of memory", -2, NULL);
    sx_ctx *ctx = thread_ctx();
    uint64_t packed_len = 0;
    uint32_t n = 0;
    const int rc = sx_fasta_pack(ctx, image, (uint64_t)fsize, packed, &packed_len, term, term_cap, &n);
    free(image);
    if (rc == SX_E_MALFORMED) {
        free(packed);
        free(term);
        if (err) *err = MALFORMED_FILE;
        return NULL;
    }
    if (rc != 0) die("load_fasta_records", rc, ctx);
    struct fasta_records *recs = malloc(sizeof *recs);
    recs->buffer = packed;
    recs->storage = malloc((n ? n : 1) * sizeof *recs->storage);
    recs->recs = NULL;
    for (uint32_t r = 0; r < n; ++r) { /* file order; every record is put in front of the list (fasta.c:127-131) */
        struct fasta_record_impl *rec = &recs->storage[r];
        rec->name = (const char *)(packed + (r ? term[2 * r - 1] + 1 : 0));
        rec->seq = packed + term[2 * r] + 1;
        rec->seq_len = term[2 * r + 1] - term[2 * r] - 1;
        rec->no_records = r + 1;
        rec->next = recs->recs;
        recs->recs = rec;
    }
    free(term);
    return recs;
}

void free_fasta_records(struct fasta_records *file)
{
    free(file->buffer);
    free(file->storage);
    free(file);
}

uint32_t number_of_fasta_records(struct fasta_records *records) { return records->recs->no_records; }

bool lookup_fasta_record_by_name(struct fasta_records *file, const char *name, struct fasta_record *record)
{
    for (struct fasta_record_impl *rec = file->recs; rec; rec = rec->next) {
        if (strcmp(rec->name, name) == 0) {
            record->name = rec->name;
            record->seq = rec->seq;
            record->seq_len = rec->seq_len;
            return true;
        }
    }
    return false;
}

void init_fasta_iter(struct fasta_iter *iter, struct fasta_records *file) { iter->rec = file->recs; }

bool next_fasta_record(struct fasta_iter *iter, struct fasta_record *rec)
{
    if (!iter->rec) return false;
    rec->name = iter->rec->name;
    rec->seq = iter->rec->seq;
    rec->seq_len = iter->rec->seq_len;
    iter->rec = iter->rec->next;
    return true;
}

void dealloc_fasta_iter(struct fasta_iter *iter) { (void)iter; }

/* ---- batch farm: independent records, host threads per GPU ------------------------------
 * Records are dealt longest first to the device with the least work so far (LPT by length, SURVEY.md section 8e:
 * a build's time is close to linear in the record's length), and every worker thread is pinned to the CPUs of
 * its GPU's NUMA node (sx_device_numa_node reads the PCI device's node from sysfs): its staging copies and the
 * page faults of the malloc'd outputs then stay on the memory next to that GPU's PCIe root. */

struct farm_job {
    const uint8_t *const *strings;
    const size_t *lengths; /* strlen of every record (the batch entry point has them) */
    struct bwt_table **out;
    const size_t *mine; /* indices of this worker's records, in the order it builds them */
    size_t n_mine;
    bool include_reverse;
    int device;
    bool on_caller;  /* the lane runs on the thread that called the farm */
    size_t failed;   /* records of this lane that could not be built (their out[] is NULL) */
    int first_error; /* the first such record's code */
};

/* "0-63,128-191" -> cpu set; returns the number of CPUs */
static int parse_cpulist(const char *list, cpu_set_t *set)
{
    int count = 0;
    CPU_ZERO(set);
    for (const char *p = list; *p && *p != '\n';) {
        char *end;
        long a = strtol(p, &end, 10), b = a;
        if (end == p) break;
        if (*end == '-') {
            p = end + 1;
            b = strtol(p, &end, 10);
            if (end == p) break;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
            CPU_SET((int)c, set);
            ++count;
        }
        p = *end == ',' ? end + 1 : end;
    }
    return count;
}

int stralg_amd_bind_thread_to_device(int device)
{
    const int node = sx_device_numa_node(device);
    if (node < 0) return -1; /* unknown (no NUMA information): leave the thread where it is */
    char path[128], list[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    const bool ok = fgets(list, sizeof list, f) != NULL;
    fclose(f);
    cpu_set_t set;
    if (!ok || parse_cpulist(list, &set) == 0) return -1;
    return pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0 ? node : -1;
}

/* One lane of the farm.  A record that cannot be built (more letters than a remap table holds, no memory on the host or
 * the device) gets out[k] = NULL and counts as a failure; the lane goes on with its other records.  on_caller: the lane
 * runs on the thread that called the farm (its thread could not be created): that thread's CPU affinity, device and
 * context are the caller's and are put back as they were. */
static void *farm_worker(void *arg)
{
    struct farm_job *job = arg;
    cpu_set_t saved_set;
    const bool have_set = job->on_caller && pthread_getaffinity_np(pthread_self(), sizeof saved_set, &saved_set) == 0;
    const int saved_device = tls_device;
    sx_ctx *saved_ctx = NULL;
    if (job->on_caller) { /* park the caller's context: stralg_amd_set_device must not destroy it */
        saved_ctx = tls_ctx;
        tls_ctx = NULL;
    }
    (void)stralg_amd_bind_thread_to_device(job->device);
    stralg_amd_set_device(job->device);
    for (size_t k = 0; k < job->n_mine; ++k) {
        struct bwt_table *t = NULL;
        const int rc = build_complete_table_try(job->strings[job->mine[k]], job->lengths[job->mine[k]], job->include_reverse, &t);
        job->out[job->mine[k]] = rc == 0 ? t : NULL;
        if (rc != 0) {
            job->failed++;
            if (!job->first_error) job->first_error = rc;
        }
    }
    stralg_amd_release();
    if (job->on_caller) {
        tls_ctx = saved_ctx;
        tls_device = saved_device;
        ctx_key_set(saved_ctx);
        if (have_set) (void)pthread_setaffinity_np(pthread_self(), sizeof saved_set, &saved_set);
    }
    return NULL;
}

struct lpt_item {
    size_t index, length;
};

static int lpt_longest_first(const void *a, const void *b)
{
    const struct lpt_item *x = a, *y = b;
    if (x->length != y->length) return x->length > y->length ? -1 : 1;
    return x->index < y->index ? -1 : (x->index > y->index ? 1 : 0); /* ties keep the given order */
}

/* assignment[k] = lane of record k (longest processing time first); lanes have equal speed */
int stralg_amd_lpt_assign(const size_t *lengths, size_t count, int lanes, int *assignment)
{
    if (lanes <= 0 || (count && (!lengths || !assignment))) return -1;
    struct lpt_item *items = malloc((count ? count : 1) * sizeof *items);
    size_t *load = calloc((size_t)lanes, sizeof *load);
    if (!items || !load) {
        free(items);
        free(load);
        return -2;
    }
    for (size_t k = 0; k < count; ++k) items[k] = (struct lpt_item){k, lengths[k]};
    qsort(items, count, sizeof *items, lpt_longest_first);
    for (size_t k = 0; k < count; ++k) {
        int best = 0;
        for (int l = 1; l < lanes; ++l)
            if (load[l] < load[best]) best = l;
        assignment[items[k].index] = best;
        load[best] += items[k].length;
    }
    free(items);
    free(load);
    return 0;
}

/* Workers a device.  A record too short to fill the GPU leaves it idle between its launches and read-backs (a build
 * takes 0.45 ms however short the record; 2^22 symbols: 0.85 ms = 4.9 Gsuffixes/s against 46 at 2^30), and a host-buffer
 * call spends most of its time on the host (page faults of the malloc'd tables, staging copies).  Several workers a
 * device -- each with its own context, stream and workspace -- fill those gaps: resident data, one MI355X, four workers:
 * 3.3 - 3.7 times the records per second up to 2^20 symbols, 2.7 times at 2^22, 2.2 times at 2^24
 * (tools/small_concurrent.py).  By the longest record (a worker's workspace grows with it: 24 B a symbol + the tables):
 * up to 2^24 symbols 4, up to 2^26 2, longer records 1; never more workers than records a device;
 * $STRALG_AMD_FARM_WORKERS overrides (1 ... 16). */
int stralg_amd_farm_workers_per_device(const size_t *lengths, size_t count, int n_devices)
{
    if (n_devices <= 0 || count == 0) return 1;
    int w = 1;
    const char *env = getenv("STRALG_AMD_FARM_WORKERS");
    if (env && *env) {
        const long v = strtol(env, NULL, 10);
        w = v < 1 ? 1 : (v > 16 ? 16 : (int)v);
    } else {
        size_t longest = 0;
        for (size_t k = 0; k < count; ++k)
            if (lengths[k] > longest) longest = lengths[k];
        w = longest <= ((size_t)1 << 24) ? 4 : (longest <= ((size_t)1 << 26) ? 2 : 1);
    }
    const size_t per_device = (count + (size_t)n_devices - 1) / (size_t)n_devices;
    if ((size_t)w > per_device) w = (int)per_device;
    return w < 1 ? 1 : w;
}

static int build_tables_batch_lanes(const uint8_t *const *strings, size_t count, bool include_reverse, const int *devices,
                                    int n_devices, struct bwt_table **out, const size_t *lengths_in);

/* the farm over records whose lengths are known */
static int build_tables_batch_known(const uint8_t *const *strings, const size_t *lengths, size_t count, bool include_reverse,
                                    const int *devices, int n_devices, struct bwt_table **out)
{
    /* every device `workers` times in the list of lanes, worker-major (lane w * n_devices + d is worker w of device d):
     * LPT deals the records over all of them */
    const int workers = stralg_amd_farm_workers_per_device(lengths, count, n_devices);
    int *lanes = malloc((size_t)n_devices * (size_t)workers * sizeof *lanes);
    if (!lanes) return -2;
    for (int w = 0; w < workers; ++w)
        for (int d = 0; d < n_devices; ++d) lanes[w * n_devices + d] = devices[d];
    const int rc = build_tables_batch_lanes(strings, count, include_reverse, lanes, n_devices * workers, out, lengths);
    free(lanes);
    return rc;
}

int stralg_amd_build_tables_batch(const uint8_t *const *strings, size_t count, bool include_reverse,
                                  const int *devices, int n_devices, struct bwt_table **out)
{
    if (!strings || !out || n_devices <= 0 || !devices) return -1;
    size_t *lengths = malloc((count ? count : 1) * sizeof *lengths);
    if (!lengths) return -2;
    for (size_t k = 0; k < count; ++k) lengths[k] = strlen((const char *)strings[k]);
    const int rc = build_tables_batch_known(strings, lengths, count, include_reverse, devices, n_devices, out);
    free(lengths);
    return rc;
}

static int build_tables_batch_lanes(const uint8_t *const *strings, size_t count, bool include_reverse, const int *devices,
                                    int n_devices, struct bwt_table **out, const size_t *lengths_in)
{
    pthread_t *threads = malloc((size_t)n_devices * sizeof *threads);
    struct farm_job *jobs = malloc((size_t)n_devices * sizeof *jobs);
    size_t *lengths = malloc((count ? count : 1) * sizeof *lengths);
    int *lane = malloc((count ? count : 1) * sizeof *lane);
    size_t *order = malloc((count ? count : 1) * sizeof *order);
    if (!threads || !jobs || !lengths || !lane || !order) {
        free(threads), free(jobs), free(lengths), free(lane), free(order);
        return -2;
    }
    for (size_t k = 0; k < count; ++k) lengths[k] = lengths_in[k];
    int rc = stralg_amd_lpt_assign(lengths, count, n_devices, lane);
    if (rc == 0) {
        /* every lane's records, longest first, as consecutive runs of `order` */
        size_t at = 0;
        for (int d = 0; d < n_devices; ++d) {
            const size_t first = at;
            for (size_t k = 0; k < count; ++k)
                if (lane[k] == d) order[at++] = k;
            for (size_t i = first + 1; i < at; ++i) /* insertion sort by length, descending (lanes are short) */
                for (size_t j = i; j > first && lengths[order[j]] > lengths[order[j - 1]]; --j) {
                    const size_t t = order[j];
                    order[j] = order[j - 1], order[j - 1] = t;
                }
            jobs[d] = (struct farm_job){strings, lengths, out, order + first, at - first, include_reverse, devices[d], false, 0, 0};
        }
        /* a lane whose thread cannot be created (EAGAIN under a thread limit) is run by the caller after the
         * others have been started: every record is built either way, and only created threads are joined */
        bool *created = calloc((size_t)n_devices, sizeof *created);
        if (!created) {
            rc = -2;
        } else {
            for (int d = 0; d < n_devices; ++d) created[d] = pthread_create(&threads[d], NULL, farm_worker, &jobs[d]) == 0;
            for (int d = 0; d < n_devices; ++d)
                if (!created[d]) {
                    jobs[d].on_caller = true;
                    (void)farm_worker(&jobs[d]);
                }
            for (int d = 0; d < n_devices; ++d)
                if (created[d]) pthread_join(threads[d], NULL);
            free(created);
            /* records that could not be built: out[k] == NULL for each, their number is the (positive) return value */
            for (int d = 0; d < n_devices; ++d) rc += (int)jobs[d].failed;
        }
    }
    free(threads), free(jobs), free(lengths), free(lane), free(order);
    return rc;
}

int stralg_amd_fasta_tables_batch_ex(struct fasta_records *records, bool include_reverse, const int *devices,
                                     int n_devices, struct bwt_table **out, size_t *n_failed)
{
    if (n_failed) *n_failed = 0;
    if (!records || !out || n_devices <= 0 || !devices) return -1;
    const uint32_t n = records->recs ? records->recs->no_records : 0;
    const uint8_t **strings = malloc((n ? n : 1) * sizeof *strings);
    size_t *lengths = malloc((n ? n : 1) * sizeof *lengths);
    if (!strings || !lengths) {
        free(strings), free(lengths);
        return -2;
    }
    uint32_t k = 0;
    for (struct fasta_record_impl *rec = records->recs; rec; rec = rec->next) {
        strings[k] = rec->seq;
        /* (a sequence of 4 Gi letters and more does not fit the record's 32-bit length: fasta.c:18's field) */
        lengths[k++] = rec->seq[rec->seq_len] == 0 ? rec->seq_len : strlen((const char *)rec->seq);
    }
    const int rc = build_tables_batch_known(strings, lengths, n, include_reverse, devices, n_devices, out);
    free(strings);
    free(lengths);
    if (rc < 0) return rc;
    if (n_failed) *n_failed = (size_t)rc;
    return (int)n;
}

/* (the records that could not be built have out[k] == NULL: check every entry, or call the _ex form for their number) */
int stralg_amd_fasta_tables_batch(struct fasta_records *records, bool include_reverse, const int *devices,
                                  int n_devices, struct bwt_table **out)
{
    return stralg_amd_fasta_tables_batch_ex(records, include_reverse, devices, n_devices, out, NULL);
}
