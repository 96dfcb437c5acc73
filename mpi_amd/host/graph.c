#define _POSIX_C_SOURCE 200809L
#include "graph.h"

#include <ctype.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "loader.h"
#include "utils.h"

#define MAX_LINE 63 /* the reference reads lines into char[64] (src/graph.c:41) */

typedef struct {
    const char *p, *end;
    int64_t lineNo;
} Scan;

static int fail(char *err, int errLen, const char *msg, int64_t lineNo) {
    if (err && errLen > 0) {
        if (lineNo > 0) snprintf(err, (size_t)errLen, "%s (input line %lld)", msg, (long long)lineNo);
        else snprintf(err, (size_t)errLen, "%s", msg);
    }
    return -1;
}

static void skipSpace(Scan *s) {
    while (s->p < s->end && isspace((unsigned char)*s->p)) {
        if (*s->p == '\n') s->lineNo++;
        s->p++;
    }
}

/* Decimal integer with optional sign at b; returns the end of the number or
 * NULL when there is none. */
static const char *parseInt(const char *b, const char *e, int64_t *out) {
    while (b < e && isspace((unsigned char)*b)) ++b;
    int neg = 0;
    if (b < e && (*b == '-' || *b == '+')) neg = *b++ == '-';
    if (b >= e || !isdigit((unsigned char)*b)) return NULL;
    int64_t v = 0;
    while (b < e && isdigit((unsigned char)*b)) {
        v = v * 10 + (*b++ - '0');
        if (v > ((int64_t)1 << 40)) return NULL;
    }
    *out = neg ? -v : v;
    return b;
}

/* Section header: the vertex count, then any amount of whitespace including
 * blank lines (reference: fscanf(file, "%d\n", ...), src/gpm-par.c:270 and
 * src/graph.c:44 -- the "\n" in that format swallows every following
 * whitespace character). */
static int scanCount(Scan *s, int64_t *n) {
    skipSpace(s);
    const char *q = parseInt(s->p, s->end, n);
    if (!q) return -1;
    s->p = q;
    skipSpace(s);
    return 0;
}

/* Next line without its newline; 0 at end of input. */
static int nextLine(Scan *s, const char **b, const char **e) {
    if (s->p >= s->end) return 0;
    const char *nl = memchr(s->p, '\n', (size_t)(s->end - s->p));
    *b = s->p;
    *e = nl ? nl : s->end;
    s->p = nl ? nl + 1 : s->end;
    s->lineNo++;
    return 1;
}

typedef struct {
    int64_t n;
    int64_t nBlocks, capBlocks;
    int32_t *blockNode;
    int64_t *blockStart; /* nBlocks + 1 */
    int64_t nNbr, capNbr;
    int32_t *nbr; /* 0-based targets in listing order */
} Section;

static void sectionFree(Section *sec) {
    free(sec->blockNode);
    free(sec->blockStart);
    free(sec->nbr);
}

/* One section: "<n>", then blocks of "<v> <deg>" and deg neighbour lines,
 * until a whitespace-only line or the end of input (reference: the loops at
 * src/gpm-par.c:276-284 and src/graph.c:54-67). */
static int scanSection(Scan *s, Section *sec, char *err, int errLen) {
    memset(sec, 0, sizeof *sec);
    if (scanCount(s, &sec->n) != 0 || sec->n < 0 || sec->n > INT32_MAX - 1)
        return fail(err, errLen, "bad vertex count", s->lineNo);
    sec->capBlocks = 1024;
    sec->capNbr = 4096;
    sec->blockNode = safeMalloc((size_t)sec->capBlocks * sizeof(int32_t));
    sec->blockStart = safeMalloc((size_t)(sec->capBlocks + 1) * sizeof(int64_t));
    sec->nbr = safeMalloc((size_t)sec->capNbr * sizeof(int32_t));
    const char *b, *e;
    while (nextLine(s, &b, &e)) {
        if (isLineEmpty(b, e)) break;
        if (e - b > MAX_LINE) return fail(err, errLen, "line longer than 63 characters", s->lineNo);
        int64_t v, deg;
        const char *q = parseInt(b, e, &v);
        if (!q || !parseInt(q, e, &deg)) return fail(err, errLen, "expected '<vertex> <degree>'", s->lineNo);
        if (v < 1 || v > sec->n) return fail(err, errLen, "vertex id out of range", s->lineNo);
        if (deg < 0) return fail(err, errLen, "negative degree", s->lineNo);
        if (sec->nBlocks == sec->capBlocks) {
            sec->capBlocks *= 2;
            sec->blockNode = safeRealloc(sec->blockNode, (size_t)sec->capBlocks * sizeof(int32_t));
            sec->blockStart = safeRealloc(sec->blockStart, (size_t)(sec->capBlocks + 1) * sizeof(int64_t));
        }
        sec->blockNode[sec->nBlocks] = (int32_t)(v - 1);
        sec->blockStart[sec->nBlocks] = sec->nNbr;
        sec->nBlocks++;
        if (sec->nNbr + deg > sec->capNbr) {
            while (sec->nNbr + deg > sec->capNbr) sec->capNbr *= 2;
            sec->nbr = safeRealloc(sec->nbr, (size_t)sec->capNbr * sizeof(int32_t));
        }
        for (int64_t j = 0; j < deg; ++j) {
            int64_t w;
            if (!nextLine(s, &b, &e)) return fail(err, errLen, "input ends inside a neighbour list", s->lineNo);
            if (e - b > MAX_LINE) return fail(err, errLen, "line longer than 63 characters", s->lineNo);
            if (!parseInt(b, e, &w)) return fail(err, errLen, "expected a neighbour id", s->lineNo);
            if (w < 1 || w > sec->n) return fail(err, errLen, "neighbour id out of range", s->lineNo);
            sec->nbr[sec->nNbr++] = (int32_t)(w - 1);
        }
    }
    sec->blockStart[sec->nBlocks] = sec->nNbr;
    return 0;
}

static void buildTranspose(Graph *g) {
    int32_t n = g->nodesCount;
    g->inOffsets = safeCalloc((size_t)n + 1, sizeof(int64_t));
    g->inEdges = safeMalloc((size_t)g->edgesCount * sizeof(int32_t));
    for (int64_t i = 0; i < g->edgesCount; ++i) g->inOffsets[g->outEdges[i] + 1]++;
    for (int32_t v = 0; v < n; ++v) g->inOffsets[v + 1] += g->inOffsets[v];
    int64_t *fill = safeMalloc((size_t)n * sizeof(int64_t));
    memcpy(fill, g->inOffsets, (size_t)n * sizeof(int64_t));
    /* sources ascending => every in-list comes out ascending */
    for (int32_t v = 0; v < n; ++v)
        for (int64_t i = g->outOffsets[v]; i < g->outOffsets[v + 1]; ++i) g->inEdges[fill[g->outEdges[i]]++] = v;
    free(fill);
}

static void classify(Graph *g) {
    int32_t n = g->nodesCount;
    g->simple = 1;
    for (int32_t v = 0; v < n && g->simple; ++v)
        for (int64_t i = g->outOffsets[v]; i < g->outOffsets[v + 1]; ++i) {
            if (g->outEdges[i] == v || (i > g->outOffsets[v] && g->outEdges[i] == g->outEdges[i - 1])) {
                g->simple = 0;
                break;
            }
        }
    g->symmetric = 1;
    for (int32_t v = 0; v < n && g->symmetric; ++v) {
        const int32_t *a = g->outEdges + g->outOffsets[v], *b = g->inEdges + g->inOffsets[v];
        int64_t na = g->outOffsets[v + 1] - g->outOffsets[v], nb = g->inOffsets[v + 1] - g->inOffsets[v];
        int64_t i = 0, j = 0;
        while (i < na && j < nb) { /* equal as sets: runs of duplicates collapse */
            if (a[i] != b[j]) break;
            int32_t x = a[i];
            while (i < na && a[i] == x) ++i;
            while (j < nb && b[j] == x) ++j;
        }
        if (i != na || j != nb) g->symmetric = 0;
    }
}

/* CSR from listing blocks; a vertex may own at most one block. */
static Graph *graphFromSection(const Section *sec, char *err, int errLen) {
    int32_t n = (int32_t)sec->n;
    Graph *g = safeCalloc(1, sizeof *g);
    g->nodesCount = n;
    g->edgesCount = sec->nNbr;
    g->outOffsets = safeCalloc((size_t)n + 1, sizeof(int64_t));
    g->outEdges = safeMalloc((size_t)sec->nNbr * sizeof(int32_t));
    char *seen = safeCalloc((size_t)n + 1, 1);
    for (int64_t b = 0; b < sec->nBlocks; ++b) {
        int32_t v = sec->blockNode[b];
        if (seen[v]) {
            free(seen);
            freeGraph(g);
            fail(err, errLen, "a data vertex is listed twice", 0);
            return NULL;
        }
        seen[v] = 1;
        g->outOffsets[v + 1] = sec->blockStart[b + 1] - sec->blockStart[b];
    }
    free(seen);
    for (int32_t v = 0; v < n; ++v) g->outOffsets[v + 1] += g->outOffsets[v];
    for (int64_t b = 0; b < sec->nBlocks; ++b) {
        int32_t v = sec->blockNode[b];
        int64_t len = sec->blockStart[b + 1] - sec->blockStart[b];
        int32_t *dst = g->outEdges + g->outOffsets[v];
        memcpy(dst, sec->nbr + sec->blockStart[b], (size_t)len * sizeof(int32_t));
        sortIds(dst, len);
    }
    buildTranspose(g);
    classify(g);
    return g;
}

static Pattern *patternFromSection(const Section *sec, char *err, int errLen) {
    if (sec->n < 1 || sec->n > MAX_MATCH_SIZE) {
        fail(err, errLen, "pattern must have 1..10 vertices", 0);
        return NULL;
    }
    Pattern *p = safeCalloc(1, sizeof *p);
    p->nodesCount = (int32_t)sec->n;
    for (int v = 0; v < MAX_MATCH_SIZE; ++v) p->labels[v] = -1;
    /* a repeated block replaces the earlier one (reference: src/graph.c:57,66) */
    for (int64_t b = 0; b < sec->nBlocks; ++b) {
        int32_t v = sec->blockNode[b];
        int64_t len = sec->blockStart[b + 1] - sec->blockStart[b];
        free(p->outEdges[v]);
        p->outEdges[v] = safeMalloc((size_t)len * sizeof(int32_t));
        memcpy(p->outEdges[v], sec->nbr + sec->blockStart[b], (size_t)len * sizeof(int32_t));
        p->outDegrees[v] = (int32_t)len;
    }
    for (int v = 0; v < p->nodesCount; ++v)
        for (int j = 0; j < p->outDegrees[v]; ++j) p->inDegrees[p->outEdges[v][j]]++;
    int32_t fill[MAX_MATCH_SIZE] = {0};
    for (int v = 0; v < p->nodesCount; ++v) p->inEdges[v] = safeMalloc((size_t)p->inDegrees[v] * sizeof(int32_t));
    for (int v = 0; v < p->nodesCount; ++v) /* reference: src/graph.c:77-82 */
        for (int j = 0; j < p->outDegrees[v]; ++j) {
            int t = p->outEdges[v][j];
            p->inEdges[t][fill[t]++] = v;
        }
    return p;
}

/* ---- side-car cache of the parsed data graph ------------------------------
 *
 * With GPM_CSR_CACHE=1 the CSR of the data-graph section is kept next to the
 * input as "<input>.gpmcsr" (or under $GPM_CSR_CACHE_DIR) and read back on the
 * next run when the input file still has the same size and modification time;
 * the pattern and the labels trailer are always parsed from the text.  The
 * benchmark runs the same multi-gigabyte input at 1, 2, 4 and 8 GPUs. */

typedef struct {
    char magic[8]; /* "GPMCSR1\0" */
    int64_t inputSize, inputMtimeSec, inputMtimeNsec;
    int64_t nodesCount, edgesCount;
    int64_t restOffset, restLine; /* where the pattern section starts in the text, lines before it */
    int32_t symmetric, simple;
} CacheHeader;

static int cachePath(const char *input, char *out, size_t outLen) {
    const char *on = getenv("GPM_CSR_CACHE");
    if (!on || !*on || strcmp(on, "0") == 0) return 0;
    const char *dir = getenv("GPM_CSR_CACHE_DIR");
    if (dir && *dir) {
        /* keyed on the whole path as given (two inputs with one file name must not share a side-car):
         * file name plus a hash of the path */
        const char *base = strrchr(input, '/');
        base = base ? base + 1 : input;
        uint64_t h = 1469598103934665603ull; /* FNV-1a */
        for (const char *c = input; *c; ++c) h = (h ^ (uint8_t)*c) * 1099511628211ull;
        return snprintf(out, outLen, "%s/%s.%016llx.gpmcsr", dir, base, (unsigned long long)h) < (int)outLen;
    }
    return snprintf(out, outLen, "%s.gpmcsr", input) < (int)outLen;
}

static int readAll(FILE *f, void *dst, size_t bytes) {
    char *p = dst;
    while (bytes) { /* a single fread of many GiB is fine on glibc, but keep the requests moderate */
        size_t want = bytes < ((size_t)1 << 30) ? bytes : ((size_t)1 << 30);
        if (fread(p, 1, want, f) != want) return -1;
        p += want;
        bytes -= want;
    }
    return 0;
}

static Graph *cacheLoad(const char *path, const struct stat *in, int64_t *restOffset, int64_t *restLine) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    CacheHeader h;
    Graph *g = NULL;
    if (fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, "GPMCSR1", 8) == 0 && h.inputSize == (int64_t)in->st_size &&
        h.inputMtimeSec == (int64_t)in->st_mtim.tv_sec && h.inputMtimeNsec == (int64_t)in->st_mtim.tv_nsec &&
        h.nodesCount >= 0 && h.nodesCount < INT32_MAX && h.edgesCount >= 0 && h.restOffset >= 0 &&
        h.restOffset <= h.inputSize && h.restLine >= 0) {
        size_t n = (size_t)h.nodesCount, m = (size_t)h.edgesCount;
        g = safeCalloc(1, sizeof *g);
        g->nodesCount = (int32_t)n;
        g->edgesCount = (int64_t)m;
        g->symmetric = h.symmetric;
        g->simple = h.simple;
        g->outOffsets = safeMalloc((n + 1) * sizeof(int64_t));
        g->outEdges = safeMalloc(m * sizeof(int32_t));
        g->inOffsets = safeMalloc((n + 1) * sizeof(int64_t));
        g->inEdges = safeMalloc(m * sizeof(int32_t));
        int ok = readAll(f, g->outOffsets, (n + 1) * sizeof(int64_t)) == 0 && readAll(f, g->outEdges, m * sizeof(int32_t)) == 0;
        if (ok && h.symmetric && h.simple) { /* the transpose is the graph itself and is not stored */
            memcpy(g->inOffsets, g->outOffsets, (n + 1) * sizeof(int64_t));
            memcpy(g->inEdges, g->outEdges, m * sizeof(int32_t));
        } else if (ok) {
            ok = readAll(f, g->inOffsets, (n + 1) * sizeof(int64_t)) == 0 && readAll(f, g->inEdges, m * sizeof(int32_t)) == 0;
        }
        ok = ok && g->outOffsets[0] == 0 && g->outOffsets[n] == (int64_t)m && g->inOffsets[0] == 0 &&
             g->inOffsets[n] == (int64_t)m;
        /* a side-car that was truncated, edited or written by something else must not reach the CSR
         * consumers unchecked: offsets ascend, ids are vertices (what the text parser guarantees) */
        for (size_t v = 0; ok && v < n; ++v)
            ok = g->outOffsets[v] <= g->outOffsets[v + 1] && g->inOffsets[v] <= g->inOffsets[v + 1];
        for (size_t e = 0; ok && e < m; ++e)
            ok = (uint32_t)g->outEdges[e] < (uint32_t)n && (uint32_t)g->inEdges[e] < (uint32_t)n;
        if (!ok) {
            freeGraph(g);
            g = NULL;
        } else {
            *restOffset = h.restOffset;
            *restLine = h.restLine;
        }
    }
    fclose(f);
    return g;
}

static void cacheStore(const char *path, const struct stat *in, const Graph *g, int64_t restOffset, int64_t restLine) {
    char tmp[4200];
    if (snprintf(tmp, sizeof tmp, "%s.tmp%d", path, (int)getpid()) >= (int)sizeof tmp) return;
    FILE *f = fopen(tmp, "wb");
    if (!f) return; /* a cache that cannot be written is not an error */
    CacheHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "GPMCSR1", 8);
    h.inputSize = (int64_t)in->st_size;
    h.inputMtimeSec = (int64_t)in->st_mtim.tv_sec;
    h.inputMtimeNsec = (int64_t)in->st_mtim.tv_nsec;
    h.nodesCount = g->nodesCount;
    h.edgesCount = g->edgesCount;
    h.restOffset = restOffset;
    h.restLine = restLine;
    h.symmetric = g->symmetric;
    h.simple = g->simple;
    size_t n = (size_t)g->nodesCount, m = (size_t)g->edgesCount;
    int ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(g->outOffsets, sizeof(int64_t), n + 1, f) == n + 1 &&
             fwrite(g->outEdges, sizeof(int32_t), m, f) == m;
    if (ok && !(g->symmetric && g->simple))
        ok = fwrite(g->inOffsets, sizeof(int64_t), n + 1, f) == n + 1 && fwrite(g->inEdges, sizeof(int32_t), m, f) == m;
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp, path) != 0) remove(tmp);
}

static int parseInput(const char *buf, int64_t len, Graph *cached, int64_t restOffset, int64_t restLine, Graph **data,
                      Pattern **pattern, int64_t *restOut, int64_t *lineOut, char *err, int errLen);

int readInputBuffer(const char *buf, int64_t len, Graph **data, Pattern **pattern, char *err, int errLen) {
    return parseInput(buf, len, NULL, 0, 0, data, pattern, NULL, NULL, err, errLen);
}

/* Parse the whole input, or with `cached` (the data graph from the side-car
 * cache) only what follows the data section.  restOut / lineOut receive where
 * the data section ended. */
static int parseInput(const char *buf, int64_t len, Graph *cached, int64_t restOffset, int64_t restLine, Graph **data,
                      Pattern **pattern, int64_t *restOut, int64_t *lineOut, char *err, int errLen) {
    *data = NULL;
    *pattern = NULL;
    Scan s = {buf, buf + len, 0};
    Section ds, ps;
    Graph *g = cached;
    if (cached) {
        s.p = buf + restOffset;
        s.lineNo = restLine;
    } else {
        /* fast path: well-formed data sections are parsed by several threads */
        Scan t = s;
        int64_t n = 0, lines = 0;
        const char *next = NULL;
        if (scanCount(&t, &n) == 0 && n >= 0 && n <= INT32_MAX - 1 &&
            (g = loadDataSectionParallel(t.p, t.end, n, loaderThreads(), &next, &lines)) != NULL) {
            s.p = next;
            s.lineNo = t.lineNo + lines;
        }
    }
    if (!g) {
        if (scanSection(&s, &ds, err, errLen) != 0) {
            sectionFree(&ds);
            return -1;
        }
        g = graphFromSection(&ds, err, errLen);
        sectionFree(&ds);
        if (!g) return -1;
    }
    if (restOut) *restOut = (int64_t)(s.p - buf);
    if (lineOut) *lineOut = s.lineNo;
    if (scanSection(&s, &ps, err, errLen) != 0) {
        sectionFree(&ps);
        freeGraph(g);
        return -1;
    }
    Pattern *p = patternFromSection(&ps, err, errLen);
    sectionFree(&ps);
    if (!p) {
        freeGraph(g);
        return -1;
    }
    /* Optional trailer, ignored by the reference (it stops reading at the
     * blank line that ends the pattern): a line that holds the single word
     * "labels", then one label per data vertex and one per pattern vertex.
     * Any other text after the pattern is ignored as the reference ignores
     * it, and GPM_LABELS=0 makes this program ignore a labels trailer too. */
    skipSpace(&s);
    const char *labelsOff = getenv("GPM_LABELS");
    int wantLabels = !(labelsOff && strcmp(labelsOff, "0") == 0);
    if (wantLabels && s.end - s.p >= 6 && memcmp(s.p, "labels", 6) == 0 &&
        (s.end - s.p == 6 || s.p[6] == '\n' || s.p[6] == '\r' || s.p[6] == ' ' || s.p[6] == '\t')) {
        s.p += 6;
        g->labels = safeMalloc((size_t)g->nodesCount);
        int ok = 1;
        for (int32_t v = 0; v < g->nodesCount && ok; ++v) {
            int64_t x;
            const char *q;
            skipSpace(&s);
            q = parseInt(s.p, s.end, &x);
            ok = q && x >= 0 && x <= 255;
            if (ok) {
                g->labels[v] = (uint8_t)x;
                s.p = q;
            }
        }
        for (int v = 0; v < p->nodesCount && ok; ++v) {
            int64_t x;
            const char *q;
            skipSpace(&s);
            q = parseInt(s.p, s.end, &x);
            ok = q && x >= 0 && x <= 255;
            if (ok) {
                p->labels[v] = (int32_t)x;
                s.p = q;
            }
        }
        if (!ok) {
            freeGraph(g);
            freePattern(p);
            return fail(err, errLen, "bad labels trailer", s.lineNo);
        }
        p->hasLabels = 1;
    }
    *data = g;
    *pattern = p;
    return 0;
}

int readInput(const char *path, Graph **data, Pattern **pattern, char *err, int errLen) {
    /* map the file when possible: the loader's threads then page it in side by side */
    int fd = open(path, O_RDONLY);
    if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                close(fd);
                char cpath[4096];
                int caching = cachePath(path, cpath, sizeof cpath);
                int64_t rest = 0, restLine = 0;
                Graph *cached = caching ? cacheLoad(cpath, &st, &rest, &restLine) : NULL;
                int r = parseInput(m, (int64_t)st.st_size, cached, rest, restLine, data, pattern, &rest, &restLine, err,
                                   errLen);
                if (r == 0 && caching && !cached) cacheStore(cpath, &st, *data, rest, restLine);
                munmap(m, (size_t)st.st_size);
                return r;
            }
        }
        close(fd);
    }
    FILE *f = fopen(path, "rb");
    if (!f) return fail(err, errLen, "cannot open input", 0);
    fseek(f, 0, SEEK_END);
    long long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = safeMalloc((size_t)len + 1);
    size_t got = fread(buf, 1, (size_t)len, f);
    fclose(f);
    int r = readInputBuffer(buf, (int64_t)got, data, pattern, err, errLen);
    free(buf);
    return r;
}

/* Decimal text of v at p; returns the end. */
static char *putInt(char *p, int64_t v) {
    char tmp[24];
    int k = 0;
    do {
        tmp[k++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}

int writeDataSection(const char *path, int32_t n, const int64_t *outOffsets, const int32_t *outEdges) {
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    const size_t cap = (size_t)1 << 22;
    char *buf = safeMalloc(cap + 64), *p = buf;
    p = putInt(p, n);
    *p++ = '\n';
    int ok = 1;
    for (int32_t v = 0; v < n && ok; ++v) {
        int64_t b = outOffsets[v], e = outOffsets[v + 1];
        if (e == b) continue; /* a "<v> 0" block crashes the reference */
        p = putInt(p, (int64_t)v + 1);
        *p++ = ' ';
        p = putInt(p, e - b);
        *p++ = '\n';
        for (int64_t i = b; i < e; ++i) {
            p = putInt(p, (int64_t)outEdges[i] + 1);
            *p++ = '\n';
            if ((size_t)(p - buf) > cap) {
                ok = fwrite(buf, 1, (size_t)(p - buf), f) == (size_t)(p - buf);
                p = buf;
            }
        }
    }
    *p++ = '\n'; /* the blank line that ends the section */
    ok = ok && fwrite(buf, 1, (size_t)(p - buf), f) == (size_t)(p - buf);
    free(buf);
    return (fclose(f) == 0 && ok) ? 0 : -1;
}

Graph *graphFromEdges(int32_t n, const int32_t *src, const int32_t *dst, int64_t m) {
    Graph *g = safeCalloc(1, sizeof *g);
    g->nodesCount = n;
    g->edgesCount = m;
    g->outOffsets = safeCalloc((size_t)n + 1, sizeof(int64_t));
    g->outEdges = safeMalloc((size_t)m * sizeof(int32_t));
    for (int64_t e = 0; e < m; ++e) {
        if (src[e] < 0 || src[e] >= n || dst[e] < 0 || dst[e] >= n) {
            freeGraph(g);
            return NULL;
        }
        g->outOffsets[src[e] + 1]++;
    }
    for (int32_t v = 0; v < n; ++v) g->outOffsets[v + 1] += g->outOffsets[v];
    int64_t *fill = safeMalloc((size_t)n * sizeof(int64_t));
    memcpy(fill, g->outOffsets, (size_t)n * sizeof(int64_t));
    for (int64_t e = 0; e < m; ++e) g->outEdges[fill[src[e]]++] = dst[e];
    free(fill);
    for (int32_t v = 0; v < n; ++v) sortIds(g->outEdges + g->outOffsets[v], g->outOffsets[v + 1] - g->outOffsets[v]);
    buildTranspose(g);
    classify(g);
    return g;
}

Pattern *patternFromEdges(int32_t n, const int32_t *src, const int32_t *dst, int32_t m) {
    if (n < 1 || n > MAX_MATCH_SIZE) return NULL;
    Section sec;
    memset(&sec, 0, sizeof sec);
    sec.n = n;
    sec.blockNode = safeMalloc((size_t)n * sizeof(int32_t));
    sec.blockStart = safeMalloc((size_t)(n + 1) * sizeof(int64_t));
    sec.nbr = safeMalloc((size_t)m * sizeof(int32_t));
    for (int v = 0; v < n; ++v) { /* one block per source, arcs in the order given */
        sec.blockNode[sec.nBlocks] = v;
        sec.blockStart[sec.nBlocks++] = sec.nNbr;
        for (int e = 0; e < m; ++e) {
            if (src[e] < 0 || src[e] >= n || dst[e] < 0 || dst[e] >= n) {
                sectionFree(&sec);
                return NULL;
            }
            if (src[e] == v) sec.nbr[sec.nNbr++] = dst[e];
        }
    }
    sec.blockStart[sec.nBlocks] = sec.nNbr;
    Pattern *p = patternFromSection(&sec, NULL, 0);
    sectionFree(&sec);
    return p;
}

void patternSetLabels(Pattern *p, const int32_t *labels) {
    p->hasLabels = labels != NULL;
    for (int v = 0; v < p->nodesCount; ++v) p->labels[v] = labels ? labels[v] : -1;
}

void freeGraph(Graph *g) {
    if (!g) return;
    free(g->outOffsets);
    free(g->outEdges);
    free(g->inOffsets);
    free(g->inEdges);
    free(g->labels);
    free(g);
}

void freePattern(Pattern *p) {
    if (!p) return;
    for (int v = 0; v < MAX_MATCH_SIZE; ++v) {
        free(p->outEdges[v]);
        free(p->inEdges[v]);
    }
    free(p);
}

void printGraph(const Graph *g) {
    printf("Graph: %d nodes\n", g->nodesCount);
    for (int32_t v = 0; v < g->nodesCount; ++v) {
        printf("Node %d --> [%lld]:", v + 1, (long long)(g->outOffsets[v + 1] - g->outOffsets[v]));
        for (int64_t i = g->outOffsets[v]; i < g->outOffsets[v + 1]; ++i) printf(" %d", g->outEdges[i] + 1);
        printf("\nNode %d <-- [%lld]:", v + 1, (long long)(g->inOffsets[v + 1] - g->inOffsets[v]));
        for (int64_t i = g->inOffsets[v]; i < g->inOffsets[v + 1]; ++i) printf(" %d", g->inEdges[i] + 1);
        printf("\n");
    }
}
