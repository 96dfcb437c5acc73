/* Independent count of triangles, 4-cliques, 4-cycles and octahedra of a large simple undirected graph.
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
 * Octahedra ("oct": K6 minus a perfect matching, 12 edges, 48 automorphisms).  A copy has three pairs of
 * opposite vertices; it is counted at the pair (u, u') that holds its lowest vertex u.  The other four
 * vertices are higher neighbours of u that are also neighbours of u', and they form a 4-cycle: with
 * T = N+(u) ^ N(u'), the copies at (u, u') are the 4-cycles of the subgraph T induces, i.e. half the sum
 * over pairs {x, y} of T of c(c-1)/2, c = common neighbours of x and y inside T.  Per u the adjacency inside
 * N+(u) is written down as bit rows and every u' above u gets a bit mask of the members of N+(u) it is
 * adjacent to (filled by walking the lists of those members).
 *
 * Input: a binary file  int64 n, int64 m, int64 off[n + 1], int32 adj[m]
 * (symmetric, no self loops, no duplicates, 0-based; list order is free).
 * usage: clique_count <file> <3 | 4 | c4 | oct> [threads]
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
static int kk, nthreads;  /* kk: 3, 4 = cliques, -4 = 4-cycles, -8 = octahedra */
static int64_t max_plus;  /* longest N+ */
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
    /* octahedra: one mask of max_plus bits per vertex, the vertices whose mask is in use, a member list */
    const size_t mw = (size_t)(max_plus + 63) / 64;
    uint64_t *mask = NULL;
    int32_t *touched = NULL, *members = NULL;
    if (kk == -8) {
        mask = calloc((size_t)n * mw + 1, sizeof(uint64_t));
        touched = malloc((size_t)n * sizeof(int32_t));
        members = malloc((size_t)(max_plus + 1) * sizeof(int32_t));
        rows = malloc(((size_t)max_plus * mw + 1) * sizeof(uint64_t));
        if (!mask || !touched || !members || !rows) {
            fprintf(stderr, "clique_count: out of memory\n");
            exit(1);
        }
    }
    for (;;) {
        int64_t u0 = __sync_fetch_and_add(&next_root, kk == -8 ? 256 : 4096);
        if (u0 >= n) break;
        int64_t u1 = u0 + (kk == -8 ? 256 : 4096) < n ? u0 + (kk == -8 ? 256 : 4096) : n;
        for (int64_t u = u0; u < u1 && kk == -8; ++u) {
            const int64_t b = poff[u], d = poff[u + 1] - b;
            if (d < 4) continue;
            for (int64_t i = 0; i < d; ++i) pos[padj[b + i]] = (int32_t)i;
            memset(rows, 0, (size_t)d * mw * sizeof(uint64_t));
            int64_t nt = 0;
            for (int64_t i = 0; i < d; ++i) {
                const int32_t x = padj[b + i];
                for (int64_t p = off[x]; p < off[x + 1]; ++p) {
                    const int32_t t = adj[p];
                    const int32_t j = pos[t];
                    if (j >= 0) rows[(size_t)i * mw + ((size_t)j >> 6)] |= 1ull << (j & 63); /* both directions */
                    if (t == u || !lower((int32_t)u, t)) continue;
                    uint64_t *mt = mask + (size_t)t * mw;
                    int seen = 0;
                    for (size_t y = 0; y < mw; ++y) seen |= mt[y] != 0;
                    if (!seen) touched[nt++] = t;
                    mt[(size_t)i >> 6] |= 1ull << (i & 63);
                }
            }
            for (int64_t z = 0; z < nt; ++z) {
                uint64_t *mt = mask + (size_t)touched[z] * mw;
                int k = 0;
                for (size_t y = 0; y < mw; ++y) {
                    uint64_t bits = mt[y];
                    while (bits) {
                        members[k++] = (int32_t)(y * 64 + (size_t)__builtin_ctzll(bits));
                        bits &= bits - 1;
                    }
                }
                if (k >= 4) {
                    uint64_t twice = 0; /* every 4-cycle of T is found at both of its diagonals */
                    for (int p = 0; p < k; ++p)
                        for (int q = p + 1; q < k; ++q) {
                            const uint64_t *rp = rows + (size_t)members[p] * mw, *rq = rows + (size_t)members[q] * mw;
                            uint64_t c = 0;
                            for (size_t y = 0; y < mw; ++y) c += (uint64_t)__builtin_popcountll(rp[y] & rq[y] & mt[y]);
                            twice += c * (c - (c ? 1 : 0)) / 2;
                        }
                    mine += twice / 2;
                }
                for (size_t y = 0; y < mw; ++y) mt[y] = 0;
            }
            for (int64_t i = 0; i < d; ++i) pos[padj[b + i]] = -1;
        }
        if (kk == -8) {
            if ((u0 & ((1 << 16) - 1)) == 0) {
                fprintf(stderr, "clique_count: roots from %" PRId64 " of %" PRId64 "\n", u0, n);
                fflush(stderr);
            }
            continue;
        }
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
    free(mask);
    free(touched);
    free(members);
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
        fprintf(stderr, "usage: clique_count <csr file> <3|4|c4|oct> [threads]\n");
        return 2;
    }
    kk = strcmp(argv[2], "c4") == 0 ? -4 : strcmp(argv[2], "oct") == 0 ? -8 : atoi(argv[2]);
    nthreads = argc > 3 ? atoi(argv[3]) : 8;
    if ((kk != 3 && kk != 4 && kk != -4 && kk != -8) || nthreads < 1 || nthreads > 256) {
        fprintf(stderr, "clique_count: pattern must be 3, 4, c4 or oct, threads 1..256\n");
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
    for (int64_t v = 0; v < n; ++v)
        if (poff[v + 1] - poff[v] > max_plus) max_plus = poff[v + 1] - poff[v];
    pthread_t th[256];
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, worker, NULL);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    unsigned __int128 emb = total;
    for (int i = 2; i <= kk; ++i) emb *= (unsigned)i;
    if (kk == -4) emb = total * 8;
    if (kk == -8) emb = total * 48;
    printf("{\"k\": %d, \"%s\": ", kk == -4 ? 4 : kk == -8 ? 6 : kk, kk == -4 ? "cycles" : kk == -8 ? "octahedra" : "cliques");
    print_u128(total);
    printf(", \"embeddings\": ");
    print_u128(emb);
    printf(", \"vertices\": %" PRId64 ", \"arcs\": %" PRId64 "}\n", n, m);
    return 0;
}
