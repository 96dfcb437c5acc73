#define _POSIX_C_SOURCE 200809L
#include "utils.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static void *checked(void *p) {
    if (p == NULL) {
        fprintf(stderr, "ERROR: malloc returned NULL");
        exit(1);
    }
    return p;
}

void *safeMalloc(size_t size) { return checked(malloc(size ? size : 1)); }

void *safeCalloc(size_t count, size_t size) { return checked(calloc(count ? count : 1, size ? size : 1)); }

void *safeRealloc(void *p, size_t size) { return checked(realloc(p, size ? size : 1)); }

int isLineEmpty(const char *s, const char *e) {
    for (; s < e; ++s)
        if (!isspace((unsigned char)*s)) return 0;
    return 1;
}

int64_t findIndex(const int32_t *tab, int64_t len, int32_t val) {
    int64_t lo = 0, hi = len;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if (tab[mid] < val) lo = mid + 1;
        else hi = mid;
    }
    return (lo < len && tab[lo] == val) ? lo : -1;
}

int arrayContains(const int32_t *tab, int64_t len, int32_t val) { return findIndex(tab, len, val) >= 0; }

static int cmpId(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

void sortIds(int32_t *a, int64_t n) {
    if (n < 2) return;
    if (n <= 24) { /* most adjacency lists are short */
        for (int64_t i = 1; i < n; ++i) {
            int32_t v = a[i];
            int64_t j = i;
            while (j > 0 && a[j - 1] > v) {
                a[j] = a[j - 1];
                --j;
            }
            a[j] = v;
        }
        return;
    }
    int sorted = 1;
    for (int64_t i = 1; i < n && sorted; ++i) sorted = a[i - 1] <= a[i];
    if (!sorted) qsort(a, (size_t)n, sizeof *a, cmpId);
}

double wallSeconds(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

uint64_t claimChunk(volatile uint64_t *cursor) {
    return __atomic_fetch_add((uint64_t *)cursor, 1, __ATOMIC_RELAXED);
}
