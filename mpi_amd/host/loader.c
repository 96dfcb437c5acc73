/*
 * Multi-threaded loader for the data-graph section of the input format
 * (reference: preprocessGraph, src/gpm-par.c:267-286, and distribiuteGraph,
 * src/gpm-par.c:394-432, which read the file twice with fgets + sscanf).
 *
 * The format is line oriented: "<v> <deg>" opens a block, the next deg lines
 * hold one neighbour id each, a whitespace-only line ends the section.  A
 * header line is the only kind with two numbers, so every thread can classify
 * the lines of its own slice of the file; prefix sums over the per-slice
 * counts then give every header and neighbour its place in the arrays.
 *
 * This is a fast path for well-formed input only.  Anything it is not sure
 * about -- a sign, extra blanks, a long line, a degree that does not match
 * the lines that follow, a vertex listed twice, ids out of range, no blank
 * line after the section -- makes it give up, and the caller runs the
 * single-threaded parser, whose behaviour and error messages are the contract.
 */
#define _GNU_SOURCE
#include "loader.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "utils.h"

#define MAX_LINE 63

typedef struct {
    const char *b, *e;      /* slice [b, e): starts at a line start */
    const char *blank;      /* start of the first whitespace-only line, or NULL */
    int64_t nHdr, nNbr, nLines;
    int64_t hdrBase, nbrBase;
    int bad;
} Slice;

typedef struct {
    int64_t n;
    int32_t *blockNode;
    int64_t *blockStart, *blockDeg;
    int32_t *nbr;
} Arrays;

typedef struct Job {
    void (*fn)(struct Job *, int);
    int threads;
    Slice *slices;
    Arrays *arr;
    Graph *g;
    int64_t nBlocks;
    int64_t cursor; /* dynamic work counter */
    int bad;
    int notSimple, notSymmetric;
} Job;

typedef struct {
    Job *job;
    int tid;
} ThreadArg;

static void *trampoline(void *p) {
    ThreadArg *a = p;
    a->job->fn(a->job, a->tid);
    return NULL;
}

static void runParallel(Job *job, void (*fn)(Job *, int)) {
    pthread_t th[64];
    ThreadArg args[64];
    job->fn = fn;
    job->cursor = 0;
    for (int t = 1; t < job->threads; ++t) {
        args[t].job = job;
        args[t].tid = t;
        if (pthread_create(&th[t], NULL, trampoline, &args[t]) != 0) {
            th[t] = 0;
            fn(job, t); /* no thread: do its share here */
        }
    }
    fn(job, 0);
    for (int t = 1; t < job->threads; ++t)
        if (th[t]) pthread_join(th[t], NULL);
}

/* One canonical line: digits, or digits ' ' digits; returns how many numbers
 * (1 or 2), 0 for a whitespace-only line, -1 for anything else. */
static inline int parseLine(const char *b, const char *e, int64_t *x, int64_t *y) {
    const char *p = b;
    if (p == e) return 0;
    if (*p < '0' || *p > '9') {
        for (; p < e; ++p)
            if (*p != ' ' && *p != '\t' && *p != '\r' && *p != '\v' && *p != '\f') return -1;
        return 0;
    }
    int64_t v = 0;
    while (p < e && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
    *x = v;
    if (p == e) return p - b <= 10 ? 1 : -1;
    if (*p != ' ' || p - b > 10) return -1;
    const char *q = ++p;
    if (p == e || *p < '0' || *p > '9') return -1;
    v = 0;
    while (p < e && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
    *y = v;
    return (p == e && p - q <= 10) ? 2 : -1;
}

/* Pass 1: classify and count the lines of a slice, up to its first blank line. */
static void countSlice(Job *job, int tid) {
    Slice *s = &job->slices[tid];
    const char *p = s->b;
    int64_t x = 0, y = 0;
    while (p < s->e) {
        const char *nl = memchr(p, '\n', (size_t)(s->e - p));
        const char *le = nl ? nl : s->e;
        if (le - p > MAX_LINE) {
            s->bad = 1;
            return;
        }
        int k = parseLine(p, le, &x, &y);
        if (k == 0) {
            s->blank = p;
            return;
        }
        if (k < 0) {
            s->bad = 1;
            return;
        }
        if (k == 2) s->nHdr++;
        else s->nNbr++;
        s->nLines++;
        p = nl ? nl + 1 : s->e;
    }
}

/* Pass 2: parse the numbers into their final places. */
static void fillSlice(Job *job, int tid) {
    Slice *s = &job->slices[tid];
    Arrays *a = job->arr;
    const char *p = s->b;
    const char *stop = s->blank ? s->blank : s->e;
    int64_t h = s->hdrBase, k = s->nbrBase, x = 0, y = 0;
    while (p < stop) {
        const char *nl = memchr(p, '\n', (size_t)(stop - p));
        const char *le = nl ? nl : stop;
        if (parseLine(p, le, &x, &y) == 2) {
            if (x < 1 || x > a->n) s->bad = 1;
            a->blockNode[h] = (int32_t)(x - 1);
            a->blockStart[h] = k;
            a->blockDeg[h] = y;
            ++h;
        } else {
            if (x < 1 || x > a->n) s->bad = 1;
            a->nbr[k++] = (int32_t)(x - 1);
        }
        p = nl ? nl + 1 : stop;
    }
}

/* Copy every block to its vertex's range of the CSR and sort it; flags
 * duplicates and self loops. */
static void placeBlocks(Job *job, int tid) {
    (void)tid;
    Arrays *a = job->arr;
    Graph *g = job->g;
    int notSimple = 0;
    for (;;) {
        int64_t b0 = __atomic_fetch_add(&job->cursor, 256, __ATOMIC_RELAXED);
        if (b0 >= job->nBlocks) break;
        int64_t b1 = b0 + 256 < job->nBlocks ? b0 + 256 : job->nBlocks;
        for (int64_t b = b0; b < b1; ++b) {
            int32_t v = a->blockNode[b];
            int64_t len = a->blockStart[b + 1] - a->blockStart[b];
            int32_t *dst = g->outEdges + g->outOffsets[v];
            memcpy(dst, a->nbr + a->blockStart[b], (size_t)len * sizeof(int32_t));
            sortIds(dst, len);
            for (int64_t i = 0; i < len; ++i)
                if (dst[i] == v || (i && dst[i] == dst[i - 1])) notSimple = 1;
        }
    }
    if (notSimple) __atomic_store_n(&job->notSimple, 1, __ATOMIC_RELAXED);
}

/* Is every arc matched by its reverse arc?  (On a graph without duplicate
 * arcs that makes every in-list equal to the out-list.) */
static void checkSymmetric(Job *job, int tid) {
    (void)tid;
    Graph *g = job->g;
    const int64_t n = g->nodesCount;
    for (;;) {
        int64_t v0 = __atomic_fetch_add(&job->cursor, 4096, __ATOMIC_RELAXED);
        if (v0 >= n || __atomic_load_n(&job->notSymmetric, __ATOMIC_RELAXED)) break;
        int64_t v1 = v0 + 4096 < n ? v0 + 4096 : n;
        for (int64_t v = v0; v < v1; ++v)
            for (int64_t i = g->outOffsets[v]; i < g->outOffsets[v + 1]; ++i) {
                int32_t w = g->outEdges[i];
                if (!arrayContains(g->outEdges + g->outOffsets[w], g->outOffsets[w + 1] - g->outOffsets[w],
                                   (int32_t)v)) {
                    __atomic_store_n(&job->notSymmetric, 1, __ATOMIC_RELAXED);
                    return;
                }
            }
    }
}

/* The same question on a graph without duplicate arcs and self loops, without
 * searching: thread t owns the targets w of one range and walks all sources v
 * in ascending order; the arcs v -> w with v < w must then show up as the
 * entries of w's sorted list below w, one after the other.  Random accesses stay inside the thread's
 * range of targets. */
static void checkSymmetricSimple(Job *job, int tid) {
    Graph *g = job->g;
    const int64_t n = g->nodesCount, m = g->edgesCount;
    const int64_t *off = g->outOffsets;
    const int32_t *adj = g->outEdges;
    /* target ranges of equal arc counts */
    int64_t lo = 0, hi = n;
    {
        int64_t a = 0, b = n; /* first w with off[w] >= m * tid / threads */
        const int64_t want = m / job->threads * tid;
        while (a < b) {
            int64_t mid = a + (b - a) / 2;
            if (off[mid] < want) a = mid + 1;
            else b = mid;
        }
        lo = tid == 0 ? 0 : a;
        a = 0, b = n;
        const int64_t want2 = m / job->threads * (tid + 1);
        while (a < b) {
            int64_t mid = a + (b - a) / 2;
            if (off[mid] < want2) a = mid + 1;
            else b = mid;
        }
        hi = tid == job->threads - 1 ? n : a;
    }
    if (hi <= lo) return;
    int64_t *cur = safeMalloc((size_t)(hi - lo) * sizeof(int64_t));
    for (int64_t w = lo; w < hi; ++w) cur[w - lo] = off[w];
    int bad = 0;
    for (int64_t v = 0; v < hi && !bad; ++v) { /* sources at or above hi have no upper neighbour in range */
        int64_t i = off[v], e = off[v + 1];
        const int32_t first = (int32_t)(v + 1 > lo ? v + 1 : lo);
        if (e - i > 16) { /* jump to the first target >= first */
            int64_t a = i, b = e;
            while (a < b) {
                int64_t mid = a + (b - a) / 2;
                if (adj[mid] < first) a = mid + 1;
                else b = mid;
            }
            i = a;
        }
        for (; i < e; ++i) {
            const int64_t w = adj[i];
            if (w < first) continue;
            if (w >= hi) break;
            const int64_t c = cur[w - lo]++;
            if (c >= off[w + 1] || adj[c] != (int32_t)v) {
                bad = 1;
                break;
            }
        }
        if ((v & 0xffff) == 0 && __atomic_load_n(&job->notSymmetric, __ATOMIC_RELAXED)) break;
    }
    for (int64_t w = lo; w < hi && !bad; ++w) { /* every entry below w was claimed by an arc towards w */
        const int64_t c = cur[w - lo];
        bad = c < off[w + 1] && adj[c] < (int32_t)w;
    }
    free(cur);
    if (bad) __atomic_store_n(&job->notSymmetric, 1, __ATOMIC_RELAXED);
}

static void countIn(Job *job, int tid) {
    (void)tid;
    Graph *g = job->g;
    const int64_t m = g->edgesCount;
    for (;;) {
        int64_t e0 = __atomic_fetch_add(&job->cursor, 1 << 16, __ATOMIC_RELAXED);
        if (e0 >= m) break;
        int64_t e1 = e0 + (1 << 16) < m ? e0 + (1 << 16) : m;
        for (int64_t e = e0; e < e1; ++e) __atomic_fetch_add(&g->inOffsets[g->outEdges[e] + 1], 1, __ATOMIC_RELAXED);
    }
}

/* Scatter sources into the in-lists (any order), using blockStart as the
 * per-vertex fill cursors. */
static void scatterIn(Job *job, int tid) {
    (void)tid;
    Graph *g = job->g;
    int64_t *fill = job->arr->blockStart;
    const int64_t n = g->nodesCount;
    for (;;) {
        int64_t v0 = __atomic_fetch_add(&job->cursor, 4096, __ATOMIC_RELAXED);
        if (v0 >= n) break;
        int64_t v1 = v0 + 4096 < n ? v0 + 4096 : n;
        for (int64_t v = v0; v < v1; ++v)
            for (int64_t i = g->outOffsets[v]; i < g->outOffsets[v + 1]; ++i)
                g->inEdges[__atomic_fetch_add(&fill[g->outEdges[i]], 1, __ATOMIC_RELAXED)] = (int32_t)v;
    }
}

static void sortIn(Job *job, int tid) {
    (void)tid;
    Graph *g = job->g;
    const int64_t n = g->nodesCount;
    for (;;) {
        int64_t v0 = __atomic_fetch_add(&job->cursor, 4096, __ATOMIC_RELAXED);
        if (v0 >= n) break;
        int64_t v1 = v0 + 4096 < n ? v0 + 4096 : n;
        for (int64_t v = v0; v < v1; ++v) sortIds(g->inEdges + g->inOffsets[v], g->inOffsets[v + 1] - g->inOffsets[v]);
    }
}

int loaderThreads(void) {
    const char *v = getenv("GPM_LOAD_THREADS");
    long t = (v && *v) ? atol(v) : 0;
    if (t <= 0) {
        t = sysconf(_SC_NPROCESSORS_ONLN);
        if (t > 16) t = 16;
    }
    if (t < 1) t = 1;
    if (t > 64) t = 64;
    return (int)t;
}

Graph *loadDataSectionParallel(const char *begin, const char *end, int64_t n, int threads, const char **next,
                               int64_t *lines) {
    const char *minBytes = getenv("GPM_LOAD_MIN_BYTES"); /* below this the threads do not pay off */
    if (threads < 2 || end - begin < ((minBytes && *minBytes) ? atoll(minBytes) : (1 << 16)) || n < 1) return NULL;
    Slice slices[64];
    memset(slices, 0, sizeof slices);
    const int64_t len = end - begin;
    const char *cut = begin;
    for (int t = 0; t < threads; ++t) {
        slices[t].b = cut;
        const char *want = t + 1 < threads ? begin + len * (t + 1) / threads : end;
        if (want < cut) want = cut;
        if (want < end) { /* move to the next line start */
            const char *nl = memchr(want, '\n', (size_t)(end - want));
            want = nl ? nl + 1 : end;
        }
        slices[t].e = cut = want;
    }
    Job job;
    memset(&job, 0, sizeof job);
    job.threads = threads;
    job.slices = slices;
    const int dbg = getenv("GPM_LOAD_DEBUG") != NULL;
    double t0 = wallSeconds();
#define PHASE(name)                                                                   \
    do {                                                                              \
        if (dbg) {                                                                    \
            double t1_ = wallSeconds();                                               \
            fprintf(stderr, "gpm load: %-14s %.3f s\n", name, t1_ - t0);              \
            t0 = t1_;                                                                 \
        }                                                                             \
    } while (0)
    runParallel(&job, countSlice);
    PHASE("count lines");

    /* the section ends at the first blank line; slices after it are not part of it */
    int last = -1;
    for (int t = 0; t < threads && last < 0; ++t) {
        if (slices[t].bad) return NULL;
        if (slices[t].blank) last = t;
    }
    if (last < 0) return NULL;
    int64_t nHdr = 0, nNbr = 0, nLines = 0;
    for (int t = 0; t <= last; ++t) {
        slices[t].hdrBase = nHdr;
        slices[t].nbrBase = nNbr;
        nHdr += slices[t].nHdr;
        nNbr += slices[t].nNbr;
        nLines += slices[t].nLines;
    }
    if (nHdr == 0) return NULL;
    for (int t = last + 1; t < threads; ++t) slices[t].e = slices[t].b; /* nothing to fill */

    Arrays arr;
    arr.n = n;
    arr.blockNode = safeMalloc((size_t)nHdr * sizeof(int32_t));
    arr.blockStart = safeMalloc(((size_t)(nHdr > n ? nHdr : n) + 2) * sizeof(int64_t));
    arr.blockDeg = safeMalloc((size_t)nHdr * sizeof(int64_t));
    arr.nbr = safeMalloc((size_t)nNbr * sizeof(int32_t));
    job.arr = &arr;
    job.nBlocks = nHdr;
    Graph *g = NULL;
    char *seen = NULL;
    runParallel(&job, fillSlice);
    PHASE("parse numbers");
    arr.blockStart[nHdr] = nNbr;
    int ok = arr.blockStart[0] == 0; /* the section opens with a header */
    for (int t = 0; t <= last; ++t) ok = ok && !slices[t].bad;
    for (int64_t b = 0; b < nHdr && ok; ++b) ok = arr.blockDeg[b] == arr.blockStart[b + 1] - arr.blockStart[b];
    if (!ok) goto giveUp;

    g = safeCalloc(1, sizeof *g);
    g->nodesCount = (int32_t)n;
    g->edgesCount = nNbr;
    g->outOffsets = safeCalloc((size_t)n + 1, sizeof(int64_t));
    g->outEdges = safeMalloc((size_t)nNbr * sizeof(int32_t));
    seen = safeCalloc((size_t)n + 1, 1);
    for (int64_t b = 0; b < nHdr && ok; ++b) {
        int32_t v = arr.blockNode[b];
        ok = !seen[v];
        seen[v] = 1;
        g->outOffsets[v + 1] = arr.blockDeg[b];
    }
    if (!ok) goto giveUp;
    for (int64_t v = 0; v < n; ++v) g->outOffsets[v + 1] += g->outOffsets[v];
    job.g = g;
    PHASE("offsets");
    runParallel(&job, placeBlocks);
    g->simple = !job.notSimple;
    PHASE("place + sort");
    runParallel(&job, g->simple ? checkSymmetricSimple : checkSymmetric);
    g->symmetric = !job.notSymmetric;
    PHASE("symmetry");

    g->inOffsets = safeCalloc((size_t)n + 1, sizeof(int64_t));
    g->inEdges = safeMalloc((size_t)nNbr * sizeof(int32_t));
    if (g->simple && g->symmetric) { /* the transpose is the graph itself */
        memcpy(g->inOffsets, g->outOffsets, ((size_t)n + 1) * sizeof(int64_t));
        memcpy(g->inEdges, g->outEdges, (size_t)nNbr * sizeof(int32_t));
    } else {
        runParallel(&job, countIn);
        for (int64_t v = 0; v < n; ++v) g->inOffsets[v + 1] += g->inOffsets[v];
        memcpy(arr.blockStart, g->inOffsets, (size_t)n * sizeof(int64_t));
        runParallel(&job, scatterIn);
        runParallel(&job, sortIn);
    }
    PHASE("transpose");
    free(seen);
    free(arr.blockNode);
    free(arr.blockStart);
    free(arr.blockDeg);
    free(arr.nbr);
    {
        const char *p = slices[last].blank; /* step over the blank line itself */
        const char *nl = memchr(p, '\n', (size_t)(end - p));
        *next = nl ? nl + 1 : end;
        *lines = nLines + 1;
    }
    return g;

giveUp:
    free(seen);
    if (g) freeGraph(g);
    free(arr.blockNode);
    free(arr.blockStart);
    free(arr.blockDeg);
    free(arr.nbr);
    return NULL;
}
