/*
 * sa_construction_harness.c -- timing harness for the constructors, in C, against the
 * reference-named API (include/stralg_compat.h).  It restates what
 * performance/suffix_array_construction.c:81-206 measures (the three constructors on
 * equal / DNA / ASCII strings, one line per measurement in the reference's format
 * "<Algo> <StringKind> <n> <seconds>"), parameterised instead of hard-coded:
 *
 *   sa_construction_harness [-n size] [-r reps] [-s seed] [-k equal|dna|ascii] [-t] [-d device]
 *
 * Differences from the reference harness: fixed-seed splitmix64 inputs instead of
 * rand()/time(NULL), heap buffers instead of a stack VLA (so MiB/GiB sizes work),
 * wall-clock time, an extra Msuffixes/s column, and -t to time build_complete_table.
 * Link with -lstralg_amd (or against libstralg to time the reference).
 */
#include "stralg_compat.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static uint64_t splitmix64_at(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* performance/suffix_array_construction.c:12-49: build_equal, build_random, build_random_large */
static uint8_t *build_string(const char *kind, size_t n, uint64_t seed)
{
    uint8_t *s = malloc(n + 1);
    for (size_t i = 0; i < n; ++i) {
        uint64_t r = splitmix64_at(seed, i) >> 33;
        if (!strcmp(kind, "equal")) s[i] = 'A';
        else if (!strcmp(kind, "dna")) s[i] = (uint8_t)"ACGT"[r % 4];
        else s[i] = (uint8_t)(1 + r % 127);
    }
    s[n] = 0;
    return s;
}

static void report(const char *algo, const char *kind, size_t n, double seconds)
{
    printf("%s %s %zu %f %.3f Msuffixes/s\n", algo, kind, n, seconds, (double)(n + 1) / seconds / 1e6);
}

int main(int argc, char **argv)
{
    size_t n = 65536;
    int reps = 3, tables = 0, device = 0, opt;
    uint64_t seed = 42;
    const char *kind = "dna";
    while ((opt = getopt(argc, argv, "n:r:s:k:td:")) != -1) {
        if (opt == 'n') n = strtoull(optarg, NULL, 10);
        else if (opt == 'r') reps = atoi(optarg);
        else if (opt == 's') seed = strtoull(optarg, NULL, 10);
        else if (opt == 'k') kind = optarg;
        else if (opt == 't') tables = 1;
        else if (opt == 'd') device = atoi(optarg);
        else return 2;
    }
    const char *label = !strcmp(kind, "equal") ? "Equal" : !strcmp(kind, "dna") ? "DNA" : "ASCII";
    if (stralg_amd_set_device(device) != 0) {
        fprintf(stderr, "no such GPU: %d\n", device);
        return 1;
    }
    uint8_t *s = build_string(kind, n, seed);
    uint8_t *remapped = malloc(n + 1);
    uint32_t alphabet_size = remap_string(remapped, s);

    for (int r = 0; r < reps; ++r) {
        double t0 = now();
        struct suffix_array *a = skew_sa_construction(s);
        double t1 = now();
        struct suffix_array *b = sa_is_construction(remapped, alphabet_size);
        double t2 = now();
        struct suffix_array *c = sa_is_mem_construction(remapped, alphabet_size);
        double t3 = now();
        report("Skew", label, n, t1 - t0);
        report("SA-IS", label, n, t2 - t1);
        report("SA-IS-MEM", label, n, t3 - t2);
        /* the three constructors return one array (tests/stralg/match_test.c:479,517,539) */
        if (memcmp(a->array, b->array, (n + 1) * sizeof(uint32_t)) != 0 ||
            memcmp(b->array, c->array, (n + 1) * sizeof(uint32_t)) != 0 || b->array[0] != n) {
            fprintf(stderr, "constructors disagree\n");
            return 1;
        }
        free_suffix_array(a);
        free_suffix_array(b);
        free_suffix_array(c);
        if (tables) {
            double t4 = now();
            struct bwt_table *tab = build_complete_table(s, true);
            double t5 = now();
            report("BWT-tables", label, n, t5 - t4);
            /* O(a, n+1) of bwt.h:49 = symbol counts: the rows must add up to n + 1 */
            uint64_t sum = 0;
            for (uint32_t x = 0; x < tab->remap_table->alphabet_size; ++x) sum += tab->o_indices[n + 1][x];
            if (sum != n + 1 || tab->c_table[0] != 0 || tab->sa->array[0] != n) {
                fprintf(stderr, "table check failed\n");
                return 1;
            }
            completely_free_bwt_table(tab);
        }
    }
    free(remapped);
    free(s);
    stralg_amd_release();
    return 0;
}
