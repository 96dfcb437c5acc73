/* Independent count of triangles, 4-cliques and 4-cycles of a large simple undirected graph.
 *
 * TEST INFRASTRUCTURE, like the rest of oracle/: nothing on the product path may
 * link or run it.  It exists to pin the counts that the benchmark prints at sizes
 * the reference (and its restatement gpm_oracle.c, which enumerates one match at
 * a time like the reference's exploreMatch, reference: src/gpm-par.c:179-243)
 * cannot reach: it shares no code and no data layout with the HIP library.
 *
 * What is counted.  The reference reports every injective map of the pattern
 * into the data graph that keeps the pattern's arcs (reference:
 * checkNodeMatches, src/gpm-par.c:88-112), so a k-clique pattern given with all
 * its k(k-1) arcs is found k! times per k-clique of a symmetric data graph.
 * This program counts the k-cliques themselves and prints both numbers.
 *
 * Method (Chiba-Nishizeki style).  Vertices are ordered by (degree, id); every
 * edge is directed from its lower to its higher end, N+(u) are u's higher
 * neighbours.  A k-clique is counted at its lowest vertex u: the subgraph that
 * N+(u) induces is written down as one bit row per member (row i = the members
 * above member i adjacent to it), triangles are the set bits, 4-cliques the sum
 * over set bits j of row i of |row_i AND row_j|.  Roots are spread over threads.
 *
 * 4-cycles (k = -4 on the command line: "c4").  A 4-cycle x - v - y - v' - x has one highest vertex x in the
 * same order and one vertex y opposite to it; with c(x, y) = number of common neighbours of x and y that lie
 * below x, the cycles with highest vertex x number the sum over y below x of c(c-1)/2.  The pattern with its
 * 8 arcs has 8 automorphisms, so the reference finds each cycle 8 times.  Per root x one array of counters
 * indexed by y is filled by walking the lists of x's lower neighbours, summed and cleared the same way.
 *
 * Input: a binary file  int64 n, int64 m, int64 off[n + 1], int32 adj[m]
 * (symmetric, no self loops, no duplicates, 0-based; list order is free).
 * usage: clique_count <file> <3 | 4 | c4> [threads]
 */
#include <inttypes.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int64_t n, m;
static int64_t *off;
static int32_t *adj;
static int64_t *poff;  /* N+ of every vertex: padj[poff[v] .. poff[v + 1]) */
static int32_t *padj;
static int kk, nthreads;
static volatile int64_t next_root;
static pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
static unsigned __int128 total;

static inline int lower(int32_t a, int32_t b) { /* a before b in (degree, id) order */
    const int64_t da = off[a + 1] - off[a], db = off[b + 1] - off[b];
    return da < db || (da == db && a < b);
}

static void *worker(void *arg) {
    (void)arg;
    int32_t *pos = malloc((size_t)n * sizeof(int32_t)); /* position in N+(u), -1 outside */
    if (!pos) {
        fprintf(stderr, "clique_count: out of memory\n");
        exit(1);
    }
    memset(pos, 0xff, (size_t)n * sizeof(int32_t));
    uint64_t *rows = NULL;
    size_t rows_cap = 0;
    unsigned __int128 mine = 0;
    for (;;) {
        int64_t u0 = __sync_fetch_and_add(&next_root, 4096);
        if (u0 >= n) break;
        int64_t u1 = u0 + 4096 < n ? u0 + 4096 : n;
        for (int64_t u = u0; u < u1 && kk == -4; ++u) {  /* 4-cycles: u is the highest vertex, pos[] counts */
            for (int64_t p = off[u]; p < off[u + 1]; ++p) {
                const int32_t v = adj[p];
                if (!lower(v, (int32_t)u)) continue;
                for (int64_t q = off[v]; q < off[v + 1]; ++q) {
                    const int32_t y = adj[q];
                    if (y != u && lower(y, (int32_t)u)) mine += (unsigned)(++pos[y]);  /* -1 + 1 = 0 first: old values add up to c(c-1)/2 */
                }
            }
            for (int64_t p = off[u]; p < off[u + 1]; ++p) {
                const int32_t v = adj[p];
                if (!lower(v, (int32_t)u)) continue;
                for (int64_t q = off[v]; q < off[v + 1]; ++q) pos[adj[q]] = -1;
            }
        }
        for (int64_t u = u0; u < u1 && kk != -4; ++u) {
            const int64_t b = poff[u], d = poff[u + 1] - b;
            if (d < kk - 1) continue;
            for (int64_t i = 0; i < d; ++i) pos[padj[b + i]] = (int32_t)i;
            if (kk == 3) {
                for (int64_t i = 0; i < d; ++i) {
                    const int32_t v = padj[b + i];
                    for (int64_t p = poff[v]; p < poff[v + 1]; ++p) mine += pos[padj[p]] >= 0;
                }
            } else {
                const size_t w = (size_t)(d + 63) / 64;
                if ((size_t)d * w > rows_cap) {
                    rows_cap = (size_t)d * w;
                    free(rows);
                    rows = malloc(rows_cap * sizeof(uint64_t));
                    if (!rows) {
                        fprintf(stderr, "clique_count: out of memory\n");
                        exit(1);
                    }
                }
                memset(rows, 0, (size_t)d * w * sizeof(uint64_t));
                for (int64_t i = 0; i < d; ++i) {
                    const int32_t v = padj[b + i];
                    for (int64_t p = poff[v]; p < poff[v + 1]; ++p) {
                        const int32_t j = pos[padj[p]];
                        if (j >= 0) rows[(size_t)i * w + ((size_t)j >> 6)] |= 1ull << (j & 63);
                    }
                }
                for (int64_t i = 0; i < d; ++i) {
                    const uint64_t *ri = rows + (size_t)i * w;
                    for (size_t x = 0; x < w; ++x) {
                        uint64_t bits = ri[x];
                        while (bits) {
                            const size_t j = x * 64 + (size_t)__builtin_ctzll(bits);
                            bits &= bits - 1;
                            const uint64_t *rj = rows + j * w;
                            uint64_t c = 0;
                            for (size_t y = 0; y < w; ++y) c += (uint64_t)__builtin_popcountll(ri[y] & rj[y]);
                            mine += c;
                        }
                    }
                }
            }
            for (int64_t i = 0; i < d; ++i) pos[padj[b + i]] = -1;
        }
        if ((u0 & ((1 << 20) - 1)) == 0) {
            fprintf(stderr, "clique_count: roots from %" PRId64 " of %" PRId64 "\n", u0, n);
            fflush(stderr);
        }
    }
    free(rows);
    free(pos);
    pthread_mutex_lock(&lock);
    total += mine;
    pthread_mutex_unlock(&lock);
    return NULL;
}

static void print_u128(unsigned __int128 v) {
    char buf[48];
    int at = 47;
    buf[at] = 0;
    do {
        buf[--at] = (char)('0' + (int)(v % 10));
        v /= 10;
    } while (v);
    fputs(buf + at, stdout);
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: clique_count <csr file> <3|4|c4> [threads]\n");
        return 2;
    }
    kk = strcmp(argv[2], "c4") == 0 ? -4 : atoi(argv[2]);
    nthreads = argc > 3 ? atoi(argv[3]) : 8;
    if ((kk != 3 && kk != 4 && kk != -4) || nthreads < 1 || nthreads > 256) {
        fprintf(stderr, "clique_count: pattern must be 3, 4 or c4, threads 1..256\n");
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(&n, 8, 1, f) != 1 || fread(&m, 8, 1, f) != 1 || n < 0 || m < 0) {
        fprintf(stderr, "clique_count: cannot read %s\n", argv[1]);
        return 1;
    }
    off = malloc((size_t)(n + 1) * 8);
    adj = malloc((size_t)(m ? m : 1) * 4);
    if (!off || !adj || fread(off, 8, (size_t)(n + 1), f) != (size_t)(n + 1) || fread(adj, 4, (size_t)m, f) != (size_t)m) {
        fprintf(stderr, "clique_count: short file %s\n", argv[1]);
        return 1;
    }
    fclose(f);
    poff = calloc((size_t)(n + 1), 8);
    if (!poff) return 1;
    for (int64_t v = 0; v < n; ++v) {
        int64_t c = 0;
        for (int64_t p = off[v]; p < off[v + 1]; ++p) c += lower((int32_t)v, adj[p]);
        poff[v + 1] = poff[v] + c;
    }
    padj = malloc((size_t)(poff[n] ? poff[n] : 1) * 4);
    if (!padj) return 1;
    for (int64_t v = 0; v < n; ++v) {
        int64_t at = poff[v];
        for (int64_t p = off[v]; p < off[v + 1]; ++p)
            if (lower((int32_t)v, adj[p])) padj[at++] = adj[p];
    }
    pthread_t th[256];
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, worker, NULL);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    unsigned __int128 emb = total;
    for (int i = 2; i <= kk; ++i) emb *= (unsigned)i;
    if (kk == -4) emb = total * 8;
    printf("{\"k\": %d, \"%s\": ", kk == -4 ? 4 : kk, kk == -4 ? "cycles" : "cliques");
    print_u128(total);
    printf(", \"embeddings\": ");
    print_u128(emb);
    printf(", \"vertices\": %" PRId64 ", \"arcs\": %" PRId64 "}\n", n, m);
    return 0;
}
