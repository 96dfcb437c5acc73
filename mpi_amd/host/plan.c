#include "plan.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "utils.h"

static int planFail(char *err, int errLen, const char *msg) {
    if (err && errLen > 0) snprintf(err, (size_t)errLen, "%s", msg);
    return -1;
}

/* Preorder numbering from vertex v (0-based): out-neighbours in listing order
 * first, then in-neighbours in ascending source order; a vertex first reached
 * along an in-edge gets a negative number (reference: dfs, src/graph.c:7-22). */
static int numberFrom(const Pattern *p, int v, int parent, int id, int *num, int32_t *parents, int viaInEdge) {
    num[v] = viaInEdge ? -id : id;
    parents[v] = parent;
    for (int j = 0; j < p->outDegrees[v]; ++j) {
        int w = p->outEdges[v][j];
        if (num[w] == 0) id = numberFrom(p, w, v, id + 1, num, parents, 0);
    }
    for (int j = 0; j < p->inDegrees[v]; ++j) {
        int w = p->inEdges[v][j];
        if (num[w] == 0) id = numberFrom(p, w, v, id + 1, num, parents, 1);
    }
    return id;
}

int makeGraphOrdering(const Pattern *p, int32_t ordering[MAX_MATCH_SIZE + 1], int32_t parents[MAX_MATCH_SIZE + 1]) {
    int num[MAX_MATCH_SIZE] = {0};
    int32_t par0[MAX_MATCH_SIZE];
    memset(ordering, 0, (MAX_MATCH_SIZE + 1) * sizeof(int32_t));
    memset(parents, 0, (MAX_MATCH_SIZE + 1) * sizeof(int32_t));
    int reached = numberFrom(p, 0, -1, 1, num, par0, 0);
    for (int v = 0; v < p->nodesCount; ++v) {
        if (num[v] == 0) continue;
        /* reference: src/graph.c:28-34, ids shifted back to 1-based */
        ordering[abs(num[v])] = num[v] > 0 ? v + 1 : -(v + 1);
        parents[v + 1] = par0[v] < 0 ? -1 : par0[v] + 1;
    }
    return reached == p->nodesCount ? 0 : -1;
}

/* ---- automorphisms and symmetry breaking ----------------------------- */

typedef struct {
    int P;
    unsigned char adj[MAX_MATCH_SIZE][MAX_MATCH_SIZE]; /* arc a->b, a != b, duplicates collapsed */
    int32_t label[MAX_MATCH_SIZE];
    int8_t *tab; /* n * P */
    uint64_t n, cap;
} AutSearch;

static void autRec(AutSearch *s, int i, int8_t *perm, unsigned used) {
    int P = s->P;
    if (i == P) {
        if (s->n == s->cap) {
            s->cap = s->cap ? s->cap * 2 : 64;
            s->tab = safeRealloc(s->tab, (size_t)s->cap * (size_t)P);
        }
        memcpy(s->tab + s->n * (uint64_t)P, perm, (size_t)P);
        s->n++;
        return;
    }
    for (int c = 0; c < P; ++c) {
        if (used >> c & 1u) continue;
        if (s->label[c] != s->label[i]) continue;
        int ok = 1;
        for (int j = 0; j < i && ok; ++j)
            ok = s->adj[i][j] == s->adj[c][perm[j]] && s->adj[j][i] == s->adj[perm[j]][c];
        if (!ok) continue;
        perm[i] = (int8_t)c;
        autRec(s, i + 1, perm, used | 1u << c);
    }
}

/* Ordering conditions lt[a][b] (bind m(a) < m(b)) that keep exactly one member
 * of every class of automorphic matches: repeatedly take the earliest-bound
 * vertex v that the remaining automorphisms still move, require m(v) to be
 * the smallest over v's orbit, and keep only the automorphisms that fix v. */
static void breakSymmetry(const AutSearch *s, const int32_t *order, unsigned char lt[MAX_MATCH_SIZE][MAX_MATCH_SIZE]) {
    int P = s->P;
    uint64_t n = s->n;
    unsigned char *alive = safeMalloc((size_t)n);
    memset(alive, 1, (size_t)n);
    memset(lt, 0, MAX_MATCH_SIZE * MAX_MATCH_SIZE);
    for (;;) {
        int chosen = -1;
        unsigned orbit = 0;
        for (int k = 0; k < P && chosen < 0; ++k) {
            int v = order[k];
            unsigned o = 0;
            for (uint64_t i = 0; i < n; ++i)
                if (alive[i]) o |= 1u << s->tab[i * (uint64_t)P + (uint64_t)v];
            if (o != 1u << v) {
                chosen = v;
                orbit = o;
            }
        }
        if (chosen < 0) break;
        for (int u = 0; u < P; ++u)
            if (u != chosen && (orbit >> u & 1u)) lt[chosen][u] = 1;
        for (uint64_t i = 0; i < n; ++i)
            if (alive[i] && s->tab[i * (uint64_t)P + (uint64_t)chosen] != chosen) alive[i] = 0;
    }
    free(alive);
}

/* "3,1,2" (1-based pattern ids) -> 0-based order; 1 when text is a permutation of 1..P. */
static int parseOrder(const char *text, int P, int *order) {
    if (!text || !*text || !(text[0] >= '0' && text[0] <= '9')) return 0;
    unsigned seen = 0;
    int k = 0;
    const char *c = text;
    while (*c) {
        char *end;
        long v = strtol(c, &end, 10);
        if (end == c || v < 1 || v > P || (seen >> v & 1u) || k >= P) return 0;
        seen |= 1u << v;
        order[k++] = (int)v - 1;
        c = end;
        if (*c == ',') ++c;
    }
    return k == P;
}

static int buildPlanOrdered(const Pattern *p, int flags, const int *forced, gpm_plan *plan, char *err, int errLen);

int buildPlan(const Pattern *p, int flags, gpm_plan *plan, char *err, int errLen) {
    return buildPlanOrdered(p, flags, NULL, plan, err, errLen);
}

static int cmpDouble(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

GraphStats *graphStatsNew(int32_t n, const int64_t *outOffsets, const int64_t *inOffsets) {
    GraphStats *s = safeCalloc(1, sizeof *s);
    double total = 0;
    double *deg = safeMalloc(((size_t)n + 1) * sizeof(double));
    for (int32_t v = 0; v < n; ++v) {
        double d = (double)(outOffsets[v + 1] - outOffsets[v]);
        d += inOffsets ? (double)(inOffsets[v + 1] - inOffsets[v]) : d;
        deg[v] = 0.5 * d; /* arcs of a symmetric graph counted once per end */
        total += deg[v];
    }
    double cap = 1;
    while (cap * cap < total) cap *= 1.0625; /* about sqrt(total) without libm */
    for (int32_t v = 0; v < n; ++v)
        if (deg[v] > cap) deg[v] = cap;
    qsort(deg, (size_t)n, sizeof(double), cmpDouble);
    int32_t nv = 0;
    for (int32_t v = 0; v < n; ++v)
        if (v == 0 || deg[v] != deg[v - 1]) nv++;
    s->nValues = nv;
    s->value = safeMalloc(((size_t)nv + 1) * sizeof(double));
    s->count = safeCalloc((size_t)nv + 1, sizeof(double));
    int32_t i = -1;
    for (int32_t v = 0; v < n; ++v) {
        if (v == 0 || deg[v] != deg[v - 1]) s->value[++i] = deg[v];
        s->count[i] += 1;
    }
    free(deg);
    for (int j = 0; j < MAX_MATCH_SIZE; ++j) {
        s->cdf[j] = safeMalloc(((size_t)nv + 1) * sizeof(double));
        double run = 0;
        for (int32_t k = 0; k < nv; ++k) {
            double pw = 1;
            for (int e = 0; e < j; ++e) pw *= s->value[k];
            run += s->count[k] * pw;
            s->cdf[j][k] = run;
        }
        s->moments[j] = run;
    }
    return s;
}

void graphStatsFree(GraphStats *s) {
    if (!s) return;
    for (int j = 0; j < MAX_MATCH_SIZE; ++j) free(s->cdf[j]);
    free(s->value);
    free(s->count);
    free(s);
}

static uint64_t rngNext(uint64_t *state) { /* splitmix64 */
    uint64_t z = (*state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

/* Index of a degree value drawn with probability proportional to count * value^j. */
static int32_t drawDegree(const GraphStats *st, int j, uint64_t *rng) {
    double u = (double)(rngNext(rng) >> 11) / 9007199254740992.0 * st->cdf[j][st->nValues - 1];
    int32_t lo = 0, hi = st->nValues - 1;
    while (lo < hi) {
        int32_t mid = lo + (hi - lo) / 2;
        if (st->cdf[j][mid] < u) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

#define COST_SAMPLES 256

/* Expected partial matches an order handles one by one (see buildPlanForGraph).  und[a][b]: pattern edge;
 * lt[a][b]: bound m(a) < m(b). */
static double orderCost(int P, const int *order, int nCore, unsigned char und[MAX_MATCH_SIZE][MAX_MATCH_SIZE],
                        unsigned char lt[MAX_MATCH_SIZE][MAX_MATCH_SIZE], const GraphStats *st) {
    /* the last two levels are counted per partial match of the level before them when their vertices are not
     * adjacent; otherwise only the last level is fused into the one before it */
    int handled = nCore - 1; /* partial matches of up to this many vertices are handled one by one */
    if (nCore >= 3 && !und[order[nCore - 2]][order[nCore - 1]]) handled = nCore - 2;
    double cost = 0;
    int deg[MAX_MATCH_SIZE] = {0};
    int edges = 0;
    uint64_t rng = 12345;
    for (int k = 0; k < handled; ++k) {
        int v = order[k];
        for (int j = 0; j < k; ++j)
            if (und[v][order[j]]) {
                deg[v]++;
                deg[order[j]]++;
                edges++;
            }
        if (k == 0) continue; /* single vertices: the same n for every order */
        double rows = 1;
        for (int j = 0; j <= k; ++j) rows *= st->moments[deg[order[j]]];
        for (int e = 0; e < edges; ++e) rows /= st->moments[1] > 0 ? st->moments[1] : 1;
        int bounded = 0;
        for (int a = 0; a <= k && !bounded; ++a)
            for (int b = 0; b <= k && !bounded; ++b) bounded = lt[order[a]][order[b]];
        if (bounded) {
            int ok = 0;
            for (int smp = 0; smp < COST_SAMPLES; ++smp) {
                double key[MAX_MATCH_SIZE];
                for (int j = 0; j <= k; ++j) /* degree value index, ties broken at random */
                    key[order[j]] = (double)drawDegree(st, deg[order[j]], &rng) + (double)(rngNext(&rng) >> 40) / 16777216.0;
                int holds = 1;
                for (int a = 0; a <= k && holds; ++a)
                    for (int b = 0; b <= k && holds; ++b)
                        if (lt[order[a]][order[b]] && !(key[order[a]] < key[order[b]])) holds = 0;
                ok += holds;
            }
            rows *= ((double)ok + 0.5) / (double)(COST_SAMPLES + 1);
        }
        cost += rows;
    }
    (void)P;
    return cost;
}

int buildPlanForGraph(const Pattern *p, int flags, const GraphStats *stats, gpm_plan *plan, char *err, int errLen) {
    int P = p->nodesCount;
    const char *orderEnv = getenv("GPM_PLAN_ORDER");
    int eligible = stats && stats->nValues > 0 && P >= 4 && P <= MAX_MATCH_SIZE && (flags & PLAN_SYMMETRIC_DATA) &&
                   (flags & PLAN_DEGREE_FILTER) && !(flags & PLAN_REFERENCE_ORDER) &&
                   !(orderEnv && *orderEnv && strcmp(orderEnv, "cost") != 0);
    if (!eligible) return buildPlanOrdered(p, flags, NULL, plan, err, errLen);
    if (buildPlanOrdered(p, flags, NULL, plan, err, errLen) != 0) return -1;

    unsigned char und[MAX_MATCH_SIZE][MAX_MATCH_SIZE];
    memset(und, 0, sizeof und);
    int nbrs[MAX_MATCH_SIZE] = {0};
    AutSearch as;
    memset(&as, 0, sizeof as);
    as.P = P;
    for (int v = 0; v < P; ++v) {
        as.label[v] = p->hasLabels ? p->labels[v] : -1;
        for (int j = 0; j < p->outDegrees[v]; ++j) {
            int w = p->outEdges[v][j];
            if (w != v) {
                und[v][w] = und[w][v] = 1;
                as.adj[v][w] = 1;
            }
        }
    }
    for (int v = 0; v < P; ++v)
        for (int w = 0; w < P; ++w) nbrs[v] += und[v][w];
    /* the core: everything but the leaves, which every candidate binds last (closed form) */
    int core[MAX_MATCH_SIZE], nCore = 0, leaves[MAX_MATCH_SIZE], nLeaves = 0, mixed = 0;
    for (int k = 0; k < P; ++k) {
        int v = plan->order[k];
        if (nbrs[v] == 1 && k > 0) {
            leaves[nLeaves++] = v;
        } else {
            if (nLeaves) mixed = 1; /* the default order has an inner vertex after a leaf: leave it alone */
            core[nCore++] = v;
        }
    }
    if (mixed || nCore < 4 || nCore > 8) return 0; /* keep the default plan */
    if (flags & PLAN_BREAK_SYMMETRY) {
        int8_t perm[MAX_MATCH_SIZE];
        autRec(&as, 0, perm, 0);
    }
    unsigned char lt[MAX_MATCH_SIZE][MAX_MATCH_SIZE];
    int32_t full[MAX_MATCH_SIZE];
    #define BOUNDS_OF(orderCore)                                                  \
        do {                                                                      \
            memset(lt, 0, sizeof lt);                                             \
            if (as.n > 1) {                                                       \
                for (int k_ = 0; k_ < nCore; ++k_) full[k_] = (orderCore)[k_];   \
                for (int k_ = 0; k_ < nLeaves; ++k_) full[nCore + k_] = leaves[k_]; \
                breakSymmetry(&as, full, lt);                                     \
            }                                                                     \
        } while (0)
    BOUNDS_OF(core);
    double defaultCost = orderCost(P, core, nCore, und, lt, stats);
    const int defaultClosedTail = !und[core[nCore - 2]][core[nCore - 1]];

    /* every order of the core whose prefixes are connected */
    int best[MAX_MATCH_SIZE], cur[MAX_MATCH_SIZE], used[MAX_MATCH_SIZE] = {0}, idx[MAX_MATCH_SIZE];
    double bestCost = defaultCost;
    memcpy(best, core, sizeof best);
    int depth = 0;
    idx[0] = -1;
    while (depth >= 0) {
        int placed = 0;
        for (int c = idx[depth] + 1; c < nCore; ++c) {
            int v = core[c];
            if (used[c]) continue;
            int connected = depth == 0;
            for (int j = 0; j < depth && !connected; ++j) connected = und[v][cur[j]];
            if (!connected) continue;
            idx[depth] = c;
            cur[depth] = v;
            used[c] = 1;
            placed = 1;
            break;
        }
        if (!placed) {
            --depth;
            if (depth >= 0) used[idx[depth]] = 0;
            continue;
        }
        if (depth + 1 == nCore) {
            /* The expected counts of dense partial patterns are far too low on graphs with a dense core, so a
             * candidate may not handle more levels row by row than the default order does: when the default ends
             * in two non-adjacent vertices (counted in closed form), so must the candidate. */
            if (defaultClosedTail && und[cur[nCore - 2]][cur[nCore - 1]]) {
                used[idx[depth]] = 0;
                continue;
            }
            BOUNDS_OF(cur);
            double c = orderCost(P, cur, nCore, und, lt, stats);
            if (c < bestCost) {
                bestCost = c;
                memcpy(best, cur, sizeof best);
            }
            used[idx[depth]] = 0;
        } else {
            ++depth;
            idx[depth] = -1;
        }
    }
    #undef BOUNDS_OF
    free(as.tab);
    if (getenv("GPM_PLAN_TRACE"))
        fprintf(stderr, "gpm plan: default order costs %.4g, best candidate %.4g\n", defaultCost, bestCost);
    if (!(bestCost < 0.5 * defaultCost)) return 0; /* the default order stays */
    int forced[MAX_MATCH_SIZE];
    for (int k = 0; k < nCore; ++k) forced[k] = best[k];
    for (int k = 0; k < nLeaves; ++k) forced[nCore + k] = leaves[k];
    free((void *)plan->aut);
    return buildPlanOrdered(p, flags, forced, plan, err, errLen);
}

/* forced: NULL, or a binding order (0-based pattern vertices) to use instead of the default one. */
static int buildPlanOrdered(const Pattern *p, int flags, const int *forced, gpm_plan *plan, char *err, int errLen) {
    int P = p->nodesCount;
    memset(plan, 0, sizeof *plan);
    if (P < 1 || P > MAX_MATCH_SIZE) return planFail(err, errLen, "pattern must have 1..10 vertices");
    int32_t ordering[MAX_MATCH_SIZE + 1], parents[MAX_MATCH_SIZE + 1];
    if (makeGraphOrdering(p, ordering, parents) != 0)
        return planFail(err, errLen, "pattern is not connected from vertex 1");

    /* The set of matches does not depend on the order in which the pattern
     * vertices are bound, only the work does.  Unless the reference's order is
     * asked for, leaves (vertices with a single neighbour) are bound last --
     * they multiply the partial matches without constraining anything, and at
     * the end they are counted in closed form -- and a search that would start
     * at a leaf starts at that leaf's neighbour instead.  Every other vertex
     * keeps its place in the reference's depth-first numbering. */
    const char *orderEnv = getenv("GPM_PLAN_ORDER");
    int keepReference = (flags & PLAN_REFERENCE_ORDER) || (orderEnv && strcmp(orderEnv, "reference") == 0);
    int nbrs[MAX_MATCH_SIZE] = {0}, someNbr[MAX_MATCH_SIZE];
    for (int v = 0; v < P; ++v) {
        unsigned seen = 0;
        someNbr[v] = -1;
        for (int j = 0; j < p->outDegrees[v]; ++j)
            if (p->outEdges[v][j] != v) seen |= 1u << p->outEdges[v][j];
        for (int j = 0; j < p->inDegrees[v]; ++j)
            if (p->inEdges[v][j] != v) seen |= 1u << p->inEdges[v][j];
        for (int w = 0; w < P; ++w)
            if (seen >> w & 1u) {
                nbrs[v]++;
                if (someNbr[v] < 0) someNbr[v] = w;
            }
    }
    int num[MAX_MATCH_SIZE] = {0};
    int32_t par0[MAX_MATCH_SIZE];
    int root = (!keepReference && P > 2 && nbrs[0] == 1) ? someNbr[0] : 0;

    /* Twins -- two vertices that are not adjacent, carry the same label and have exactly the same
     * arcs to and from every other vertex -- are bound last as well when the pattern has no leaves:
     * their candidates are then one common set, counted as pairs in closed form, instead of one
     * extension level per twin.  The numbering runs over the other vertices only (they must stay
     * connected) and each twin hangs off its first neighbour. */
    int twinA = -1, twinB = -1;
    if (!keepReference && P >= 4) {
        int leaves = 0;
        for (int v = 0; v < P; ++v) leaves += nbrs[v] == 1;
        unsigned char arc[MAX_MATCH_SIZE][MAX_MATCH_SIZE];
        memset(arc, 0, sizeof arc);
        for (int v = 0; v < P; ++v)
            for (int j = 0; j < p->outDegrees[v]; ++j)
                if (p->outEdges[v][j] != v) arc[v][p->outEdges[v][j]] = 1;
        for (int a = P - 1; a >= 1 && twinA < 0 && leaves == 0; --a)
            for (int b = a - 1; b >= 0 && twinA < 0; --b) {
                if (arc[a][b] || arc[b][a] || nbrs[a] == 0) continue;
                if (p->hasLabels && p->labels[a] != p->labels[b]) continue;
                int same = 1;
                for (int t = 0; t < P && same; ++t)
                    if (t != a && t != b) same = arc[a][t] == arc[b][t] && arc[t][a] == arc[t][b];
                if (!same) continue;
                /* the rest must be connected from the root the search would use */
                int r = root;
                if (r == a || r == b) r = someNbr[r];
                int tmpNum[MAX_MATCH_SIZE] = {0};
                int32_t tmpPar[MAX_MATCH_SIZE];
                tmpNum[a] = tmpNum[b] = MAX_MATCH_SIZE + 1;
                if (numberFrom(p, r, -1, 1, tmpNum, tmpPar, 0) != P - 2) continue;
                twinA = b; /* the lower pattern vertex first */
                twinB = a;
                root = r;
            }
    }
    if (twinA >= 0) {
        num[twinA] = num[twinB] = MAX_MATCH_SIZE + 1; /* not to be entered by the numbering */
        if (numberFrom(p, root, -1, 1, num, par0, 0) != P - 2)
            return planFail(err, errLen, "pattern is not connected from vertex 1");
        int id = P - 2;
        const int tw[2] = {twinA, twinB};
        for (int k = 0; k < 2; ++k) {
            int v = tw[k], parent = -1, inEdge = 0;
            for (int w = 0; w < P && parent < 0; ++w) { /* first neighbour that lists v, else that v lists */
                if (w == twinA || w == twinB) continue;
                for (int j = 0; j < p->outDegrees[w] && parent < 0; ++j)
                    if (p->outEdges[w][j] == v) parent = w;
            }
            if (parent < 0)
                for (int j = 0; j < p->outDegrees[v] && parent < 0; ++j)
                    if (p->outEdges[v][j] != v) {
                        parent = p->outEdges[v][j];
                        inEdge = 1;
                    }
            ++id;
            num[v] = inEdge ? -id : id;
            par0[v] = parent;
        }
    } else if (numberFrom(p, root, -1, 1, num, par0, 0) != P)
        return planFail(err, errLen, "pattern is not connected from vertex 1");
    int seq[MAX_MATCH_SIZE], viaIn[MAX_MATCH_SIZE];
    for (int v = 0; v < P; ++v) {
        seq[abs(num[v]) - 1] = v;
        viaIn[v] = num[v] < 0;
    }
    int levelOf[MAX_MATCH_SIZE];
    plan->n_levels = P;
    plan->n_aut = 1;
    int explicitOrder[MAX_MATCH_SIZE];
    int haveExplicit = !keepReference && (parseOrder(orderEnv, P, explicitOrder) || (forced && forced[0] >= 0));
    if (haveExplicit && forced && forced[0] >= 0) memcpy(explicitOrder, forced, (size_t)P * sizeof(int));
    if (haveExplicit) {
        /* A given binding order (GPM_PLAN_ORDER="3,1,2,..." with the pattern's 1-based ids, or the cost
         * model's choice): every vertex after the first hangs off its earliest-bound neighbour, over an
         * out-arc of that neighbour when there is one. */
        unsigned char arc[MAX_MATCH_SIZE][MAX_MATCH_SIZE];
        memset(arc, 0, sizeof arc);
        for (int v = 0; v < P; ++v)
            for (int j = 0; j < p->outDegrees[v]; ++j)
                if (p->outEdges[v][j] != v) arc[v][p->outEdges[v][j]] = 1;
        for (int k = 0; k < P; ++k) {
            int v = explicitOrder[k];
            plan->order[k] = v;
            levelOf[v] = k;
            par0[v] = -1;
            viaIn[v] = 0;
            for (int j = 0; j < k && par0[v] < 0; ++j) {
                int w = explicitOrder[j];
                if (arc[w][v]) par0[v] = w;
                else if (arc[v][w]) {
                    par0[v] = w;
                    viaIn[v] = 1;
                }
            }
            if (k > 0 && par0[v] < 0) return planFail(err, errLen, "binding order leaves a vertex without a bound neighbour");
        }
    } else {
        int k = 0;
        for (int pass = 0; pass < 2; ++pass) /* pass 0: the root and the inner vertices, pass 1: the leaves */
            for (int i = 0; i < P; ++i) {
                int v = seq[i];
                int leaf = !keepReference && v != root && nbrs[v] == 1;
                if (leaf == pass) {
                    plan->order[k] = v;
                    levelOf[v] = k++;
                }
            }
    }
    int symData = (flags & PLAN_SYMMETRIC_DATA) != 0;

    /* effective pattern arcs: self loops never constrain anything (the vertex
     * is still unbound when it is tested, reference: src/gpm-par.c:96-97) */
    AutSearch as;
    memset(&as, 0, sizeof as);
    as.P = P;
    for (int v = 0; v < P; ++v) {
        as.label[v] = p->hasLabels ? p->labels[v] : -1;
        for (int j = 0; j < p->outDegrees[v]; ++j)
            if (p->outEdges[v][j] != v) as.adj[v][p->outEdges[v][j]] = 1;
    }

    for (int k = 0; k < P; ++k) {
        int pv = plan->order[k];
        plan->label[k] = p->hasLabels ? p->labels[pv] : -1;
        int dOut = 0, dIn = 0;
        for (int t = 0; t < P; ++t) {
            dOut += as.adj[pv][t];
            dIn += as.adj[t][pv];
        }
        if (flags & PLAN_DEGREE_FILTER) {
            plan->min_out_deg[k] = dOut;
            plan->min_in_deg[k] = dIn;
        }
        if (k == 0) {
            plan->src_level[0] = -1;
            continue;
        }
        int parent = par0[pv];
        int srcDir = viaIn[pv]; /* reached through an in-edge: walk the parent's in-list */
        plan->src_level[k] = levelOf[parent];
        plan->src_dir[k] = symData ? 0 : srcDir;
        /* every bound pattern neighbour t constrains the candidate c:
         *   arc pv->t : m(t) in N_out(c)  <=>  c in N_in(m(t))
         *   arc t->pv : m(t) in N_in(c)   <=>  c in N_out(m(t))
         * (reference: src/gpm-par.c:94-110; in-lists are the exact transpose) */
        int nc = 0;
        for (int lt = 0; lt < k; ++lt) {
            int t = plan->order[lt];
            for (int dir = 0; dir < 2; ++dir) {
                int has = dir == 0 ? as.adj[t][pv] : as.adj[pv][t];
                if (!has) continue;
                int d = symData ? 0 : dir;
                if (lt == plan->src_level[k] && d == plan->src_dir[k]) continue; /* the tree edge itself */
                int dup = 0;
                for (int j = 0; j < nc; ++j) dup |= plan->cons_level[k][j] == lt && plan->cons_dir[k][j] == d;
                if (dup) continue;
                plan->cons_level[k][nc] = lt;
                plan->cons_dir[k][nc] = d;
                nc++;
            }
        }
        plan->n_cons[k] = nc;
    }

    if (flags & PLAN_BREAK_SYMMETRY) {
        int8_t perm[MAX_MATCH_SIZE];
        autRec(&as, 0, perm, 0);
        if (as.n > 1) {
            unsigned char lt[MAX_MATCH_SIZE][MAX_MATCH_SIZE], closure[MAX_MATCH_SIZE][MAX_MATCH_SIZE];
            breakSymmetry(&as, plan->order, lt);
            memcpy(closure, lt, sizeof closure);
            for (int m = 0; m < P; ++m)
                for (int a = 0; a < P; ++a)
                    for (int b = 0; b < P; ++b) closure[a][b] |= closure[a][m] & closure[m][b];
            for (int k = 1; k < P; ++k) {
                int u = plan->order[k], nl = 0;
                for (int a = 0; a < P; ++a) {
                    if (!lt[a][u]) continue;
                    int implied = 0; /* a < w < u for another direct bound w */
                    for (int w = 0; w < P; ++w) implied |= w != a && lt[w][u] && closure[a][w];
                    if (!implied) plan->lower_level[k][nl++] = levelOf[a];
                }
                plan->n_lower[k] = nl;
            }
            /* move the identity to the front */
            for (uint64_t i = 0; i < as.n; ++i) {
                int id = 1;
                for (int v = 0; v < P; ++v) id &= as.tab[i * (uint64_t)P + (uint64_t)v] == v;
                if (id && i != 0) {
                    memcpy(as.tab + i * (uint64_t)P, as.tab, (size_t)P);
                    for (int v = 0; v < P; ++v) as.tab[v] = (int8_t)v;
                }
            }
            plan->n_aut = as.n;
            plan->aut = as.tab;
            as.tab = NULL;
        }
        free(as.tab);
    }
    return 0;
}
