/*
 * gpm_oracle.c -- CPU oracle for the gpm-par match search.
 *
 * TEST INFRASTRUCTURE ONLY (see gpm_oracle.h).  Single-threaded plain C
 * restatement of the reference algorithm; every function cites the
 * reference lines it follows.  Parity pin: byte-identical output files
 * against the compiled reference at `mpirun -np 2` on all fixtures in
 * tests/golden/.
 */
#define _POSIX_C_SOURCE 200809L
#include "gpm_oracle.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#define LINE_CAP 64 /* reference: char line[64], src/graph.c:41, src/gpm-par.c:268 */

static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz);
    if (!p) {
        fprintf(stderr, "oracle: out of memory\n");
        exit(2);
    }
    return p;
}

/* reference: isLineEmpty, src/utils.c:9-18 -- a line of only whitespace
 * (including the empty string) ends a section. */
static int blank_line(const char *s) {
    for (; *s; ++s)
        if (!isspace((unsigned char)*s)) return 0;
    return 1;
}

static OGraph *graph_alloc(int n) {
    OGraph *g = xcalloc(1, sizeof *g);
    g->n = n;
    g->out = xcalloc((size_t)n + 1, sizeof(int *));
    g->in = xcalloc((size_t)n + 1, sizeof(int *));
    g->outdeg = xcalloc((size_t)n + 1, sizeof(int));
    g->indeg = xcalloc((size_t)n + 1, sizeof(int));
    return g;
}

void oracle_graph_free(OGraph *g) {
    if (!g) return;
    for (int v = 0; v <= g->n; ++v) {
        free(g->out[v]);
        free(g->in[v]);
    }
    free(g->out);
    free(g->in);
    free(g->outdeg);
    free(g->indeg);
    free(g->ordering);
    free(g->parents);
    free(g);
}

/* reference: in-list construction, src/graph.c:68-83 (pattern) and the
 * one-worker case of transpose, src/gpm-par.c:523-531 (data graph): walk
 * sources in ascending id, targets in list order, append source to the
 * target's in-list. */
static void build_in_lists(OGraph *g) {
    for (int v = 1; v <= g->n; ++v)
        g->in[v] = g->indeg[v] ? xcalloc((size_t)g->indeg[v], sizeof(int)) : NULL;
    int *fill = xcalloc((size_t)g->n + 1, sizeof(int));
    for (int v = 1; v <= g->n; ++v)
        for (int j = 0; j < g->outdeg[v]; ++j) {
            int t = g->out[v][j];
            g->in[t][fill[t]++] = v;
        }
    free(fill);
}

OGraph *oracle_graph_from_edges(int n, const int *src, const int *dst, long long m) {
    OGraph *g = graph_alloc(n);
    for (long long e = 0; e < m; ++e) {
        if (src[e] < 1 || src[e] > n || dst[e] < 1 || dst[e] > n) {
            oracle_graph_free(g);
            return NULL;
        }
        g->outdeg[src[e]]++;
        g->indeg[dst[e]]++;
    }
    int *fill = xcalloc((size_t)n + 1, sizeof(int));
    for (int v = 1; v <= n; ++v)
        g->out[v] = g->outdeg[v] ? xcalloc((size_t)g->outdeg[v], sizeof(int)) : NULL;
    for (long long e = 0; e < m; ++e) g->out[src[e]][fill[src[e]]++] = dst[e];
    free(fill);
    build_in_lists(g);
    return g;
}

/* reference: dfs, src/graph.c:7-22.  Preorder numbering; out-neighbours in
 * list order first, then in-neighbours; a vertex reached through an
 * in-edge gets a negative number. */
static int plan_dfs(OGraph *p, int v, int parent, int next_id, int *num, int reversed) {
    int id = next_id;
    num[v] = reversed ? -id : id;
    p->parents[v] = parent;
    for (int j = 0; j < p->outdeg[v]; ++j) {
        int w = p->out[v][j];
        if (num[w] == 0) id = plan_dfs(p, w, v, id + 1, num, 0);
    }
    for (int j = 0; j < p->indeg[v]; ++j) {
        int w = p->in[v][j];
        if (num[w] == 0) id = plan_dfs(p, w, v, id + 1, num, 1);
    }
    return id;
}

/* reference: makeGraphOrdering, src/graph.c:24-36. */
int oracle_make_plan(OGraph *p) {
    free(p->ordering);
    free(p->parents);
    p->ordering = xcalloc((size_t)p->n + 1, sizeof(int));
    p->parents = xcalloc((size_t)p->n + 1, sizeof(int));
    int *num = xcalloc((size_t)p->n + 1, sizeof(int));
    int reached = 0;
    if (p->n >= 1) reached = plan_dfs(p, 1, -1, 1, num, 0);
    for (int v = 1; v <= p->n; ++v) {
        if (num[v] > 0) p->ordering[num[v]] = v;
        else if (num[v] < 0) p->ordering[-num[v]] = -v;
    }
    free(num);
    return reached == p->n ? 0 : -1;
}

/* One section of the input file (reference: the read loops of
 * preprocessGraph src/gpm-par.c:270-284 and readGraph src/graph.c:44-67):
 * "<n>", then blocks "<v> <deg>" followed by deg neighbour lines, ended by
 * a whitespace-only line or EOF. */
static OGraph *read_section(FILE *f) {
    int n;
    if (fscanf(f, "%d\n", &n) != 1 || n < 0) return NULL;
    OGraph *g = graph_alloc(n);
    char line[LINE_CAP];
    while (fgets(line, sizeof line, f) && !blank_line(line)) {
        int v, deg;
        if (sscanf(line, "%d %d", &v, &deg) != 2 || v < 1 || v > n || deg < 0) {
            oracle_graph_free(g);
            return NULL;
        }
        free(g->out[v]);
        g->out[v] = xcalloc((size_t)deg, sizeof(int));
        g->outdeg[v] = deg;
        for (int j = 0; j < deg; ++j) {
            int w;
            if (!fgets(line, sizeof line, f) || sscanf(line, "%d", &w) != 1 || w < 1 || w > n) {
                oracle_graph_free(g);
                return NULL;
            }
            g->out[v][j] = w;
            g->indeg[w]++;
        }
    }
    build_in_lists(g);
    return g;
}

int oracle_parse_file(const char *path, OGraph **data, OGraph **pattern,
                      int **data_labels, int **pattern_labels) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    *data = read_section(f);
    *pattern = *data ? read_section(f) : NULL;
    if (data_labels) *data_labels = NULL;
    if (pattern_labels) *pattern_labels = NULL;
    if (!*data || !*pattern) {
        fclose(f);
        return -1;
    }
    oracle_make_plan(*pattern);
    /* optional trailer (ignored by the reference): "labels", n + P ints */
    char line[LINE_CAP];
    while (fgets(line, sizeof line, f)) {
        if (blank_line(line)) continue;
        if (strncmp(line, "labels", 6) == 0 && data_labels && pattern_labels) {
            int n = (*data)->n, P = (*pattern)->n;
            int *dl = xcalloc((size_t)n + 1, sizeof(int));
            int *pl = xcalloc((size_t)P + 1, sizeof(int));
            int ok = 1;
            for (int v = 1; v <= n && ok; ++v) ok = fscanf(f, "%d", &dl[v]) == 1;
            for (int v = 1; v <= P && ok; ++v) ok = fscanf(f, "%d", &pl[v]) == 1;
            if (!ok) {
                free(dl);
                free(pl);
                fclose(f);
                return -1;
            }
            *data_labels = dl;
            *pattern_labels = pl;
        }
        break;
    }
    fclose(f);
    return 0;
}

/* ---- search -------------------------------------------------------- */

typedef struct {
    const OGraph *g, *p;
    int fast;
    int prune;         /* fast == 2: test a vertex's label when it is bound instead of on complete matches */
    int **sout, **sin; /* sorted copies for fast membership */
    int symmetric;     /* data graph: in-list == out-list as sets for all v */
    const int *dl, *pl;
    int *rows;
    long long cap;
    FILE *out;
    OStats st;
} Search;

static int cmp_int(const void *a, const void *b) {
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* reference: arrayContains, src/utils.c:46-53 (linear scan of an unsorted
 * list); the fast variant answers the same question by binary search on a
 * sorted copy. */
static int list_has(const int *lst, const int *sorted, int len, int val, int fast) {
    if (!fast) {
        for (int i = 0; i < len; ++i)
            if (lst[i] == val) return 1;
        return 0;
    }
    int lo = 0, hi = len;
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (sorted[mid] < val) lo = mid + 1;
        else hi = mid;
    }
    return lo < len && sorted[lo] == val;
}

/* reference: checkNodeMatches, src/gpm-par.c:88-112.  m[] is indexed by
 * pattern vertex, -1 = unmatched (Match, src/gpm-par.c:28-31). */
static int candidate_ok(const Search *s, int c, int pv, const int *m) {
    for (int i = 1; i <= ORACLE_MAX_PATTERN; ++i) /* matchContains, :79-86 */
        if (m[i] == c) return 0;
    const OGraph *g = s->g, *p = s->p;
    for (int j = 0; j < p->outdeg[pv]; ++j) {
        int gv = m[p->out[pv][j]];
        if (gv > -1 && !list_has(g->out[c], s->sout ? s->sout[c] : NULL, g->outdeg[c], gv, s->fast))
            return 0;
    }
    for (int j = 0; j < p->indeg[pv]; ++j) {
        int gv = m[p->in[pv][j]];
        if (gv > -1 && !list_has(g->in[c], s->sin ? s->sin[c] : NULL, g->indeg[c], gv, s->fast))
            return 0;
    }
    return 1;
}

static void emit(Search *s, const int *m) {
    int P = s->p->n;
    if (s->dl && s->pl)
        for (int i = 1; i <= P; ++i)
            if (s->dl[m[i]] != s->pl[i]) return;
    if (s->rows && s->st.matches < s->cap)
        memcpy(s->rows + s->st.matches * P, m + 1, (size_t)P * sizeof(int));
    if (s->out) { /* writeMatch, src/gpm-par.c:71-77 */
        fprintf(s->out, "%d", m[1]);
        for (int i = 2; i <= P; ++i) fprintf(s->out, " %d", m[i]);
        fputc('\n', s->out);
    }
    s->st.matches++;
}

/* Merge-model traffic of extending one partial match of length k-1 to
 * level k: the partial match itself, the candidate source list, and one
 * pass over the list of every other matched pattern neighbour (a
 * constraint that repeats the tree edge is skipped; on a symmetric data
 * graph an edge pair p<->t counts once). */
static long long level_bytes(const Search *s, int k, int pv, int parent, int reversed, const int *m) {
    const OGraph *g = s->g, *p = s->p;
    long long b = 4LL * (k - 1);
    b += 4LL * (reversed ? g->indeg[m[parent]] : g->outdeg[m[parent]]);
    /* constraint (t, dir): candidate must lie in N_dir(m(t)); dir 0 = out */
    int seen[ORACLE_MAX_PATTERN + 1][2];
    memset(seen, 0, sizeof seen);
    seen[parent][s->symmetric ? 0 : (reversed ? 1 : 0)] = 1;
    for (int j = 0; j < p->outdeg[pv]; ++j) { /* pv -> t  =>  c in N_in(m(t)) */
        int t = p->out[pv][j];
        if (t == pv || m[t] < 0) continue;
        int d = s->symmetric ? 0 : 1;
        if (seen[t][d]) continue;
        seen[t][d] = 1;
        b += 4LL * (d ? g->indeg[m[t]] : g->outdeg[m[t]]);
    }
    for (int j = 0; j < p->indeg[pv]; ++j) { /* t -> pv  =>  c in N_out(m(t)) */
        int t = p->in[pv][j];
        if (t == pv || m[t] < 0) continue;
        if (seen[t][0]) continue;
        seen[t][0] = 1;
        b += 4LL * g->outdeg[m[t]];
    }
    return b;
}

/* reference: exploreMatch, src/gpm-par.c:179-243 (local-vertex branch;
 * the remote-fetch branch does not exist with one worker). */
static void explore(Search *s, int depth, int *m) {
    const OGraph *g = s->g, *p = s->p;
    if (depth == p->n) {
        emit(s, m);
        return;
    }
    int signed_next = p->ordering[depth + 1];
    int pv = abs(signed_next);
    int parent = p->parents[pv];
    int gv = m[parent];
    const int *cand = signed_next < 0 ? g->in[gv] : g->out[gv];
    int ncand = signed_next < 0 ? g->indeg[gv] : g->outdeg[gv];
    s->st.extended[depth]++;
    s->st.alg_bytes += level_bytes(s, depth + 1, pv, parent, signed_next < 0, m);
    for (int i = 0; i < ncand; ++i) {
        int c = cand[i];
        /* The label filter keeps a complete match iff every vertex carries its pattern vertex's label, so
         * a candidate with the wrong label can be dropped here: the surviving matches and their order
         * are those of the post-filter (the per-level statistics then count the pruned search). */
        if (s->prune && s->dl[c] != s->pl[pv]) continue;
        if (candidate_ok(s, c, pv, m)) {
            int m2[ORACLE_MAX_PATTERN + 1]; /* addNode, src/gpm-par.c:169-177 */
            memcpy(m2, m, sizeof m2);
            m2[pv] = c;
            explore(s, depth + 1, m2);
        }
    }
}

static int **sorted_copy(int **lst, const int *deg, int n) {
    int **s = xcalloc((size_t)n + 1, sizeof(int *));
    for (int v = 1; v <= n; ++v) {
        if (!deg[v]) continue;
        s[v] = xcalloc((size_t)deg[v], sizeof(int));
        memcpy(s[v], lst[v], (size_t)deg[v] * sizeof(int));
        qsort(s[v], (size_t)deg[v], sizeof(int), cmp_int);
    }
    return s;
}

static void free_lists(int **s, int n) {
    if (!s) return;
    for (int v = 0; v <= n; ++v) free(s[v]);
    free(s);
}

static int same_set(const int *a, int na, const int *b, int nb) {
    /* a, b sorted; compare as sets (duplicates collapse) */
    int i = 0, j = 0;
    while (i < na && j < nb) {
        if (a[i] != b[j]) return 0;
        int v = a[i];
        while (i < na && a[i] == v) ++i;
        while (j < nb && b[j] == v) ++j;
    }
    return i == na && j == nb;
}

long long oracle_search(const OGraph *data, const OGraph *pattern, int fast,
                        const int *data_labels, const int *pattern_labels,
                        int root_lo, int root_hi,
                        int *rows, long long cap, FILE *out, OStats *stats) {
    if (!data || !pattern || !pattern->ordering || pattern->n < 1 || pattern->n > ORACLE_MAX_PATTERN)
        return -1;
    Search s;
    memset(&s, 0, sizeof s);
    s.g = data;
    s.p = pattern;
    s.fast = fast;
    s.prune = fast == 2 && data_labels && pattern_labels;
    s.dl = data_labels;
    s.pl = pattern_labels;
    s.rows = rows;
    s.cap = cap;
    s.out = out;
    int **so = sorted_copy(data->out, data->outdeg, data->n);
    int **si = sorted_copy(data->in, data->indeg, data->n);
    s.symmetric = 1;
    for (int v = 1; v <= data->n && s.symmetric; ++v)
        s.symmetric = same_set(so[v], data->outdeg[v], si[v], data->indeg[v]);
    if (fast) {
        s.sout = so;
        s.sin = si;
    }
    if (root_lo < 1) root_lo = 1;
    if (root_hi > data->n) root_hi = data->n;
    /* reference: findMatches, src/gpm-par.c:245-252 -- every vertex is a
     * root for pattern vertex 1, ascending id. */
    for (int v = root_lo; v <= root_hi; ++v) {
        if (s.prune && data_labels[v] != pattern_labels[1]) continue;
        int m[ORACLE_MAX_PATTERN + 1];
        for (int i = 0; i <= ORACLE_MAX_PATTERN; ++i) m[i] = -1;
        m[1] = v;
        explore(&s, 1, m);
    }
    free_lists(so, data->n);
    free_lists(si, data->n);
    if (stats) *stats = s.st;
    return s.st.matches;
}

long long oracle_run_file(const char *in_path, const char *out_path, int fast) {
    OGraph *g = NULL, *p = NULL;
    int *dl = NULL, *pl = NULL;
    if (oracle_parse_file(in_path, &g, &p, &dl, &pl) != 0) return -1;
    FILE *out = fopen(out_path, "w+");
    if (!out) return -1;
    long long r = oracle_search(g, p, fast, dl, pl, 1, g->n, NULL, 0, out, NULL);
    fclose(out);
    oracle_graph_free(g);
    oracle_graph_free(p);
    free(dl);
    free(pl);
    return r;
}

#ifdef ORACLE_MAIN
#include <time.h>
int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <inFile> <outFile> [fast]\n", argv[0]);
        return 1;
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    long long r = oracle_run_file(argv[1], argv[2], argc > 3);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (r < 0) {
        fprintf(stderr, "oracle: failed on %s\n", argv[1]);
        return 1;
    }
    fprintf(stderr, "oracle: %lld matches in %.3f s\n", r,
            (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
    return 0;
}
#endif
