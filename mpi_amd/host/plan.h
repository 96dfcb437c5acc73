/*
 * Matching plan for a pattern (counterpart of the reference's
 * makeGraphOrdering, src/graph.c:7-36, plus what exploreMatch and
 * checkNodeMatches derive from the pattern at run time, src/gpm-par.c:88-112
 * and :186-206).
 */
#ifndef GPM_HOST_PLAN_H_
#define GPM_HOST_PLAN_H_

#include "../../include/gpm_hip.h"
#include "graph.h"

/* The reference's own plan arrays, 1-based like the reference keeps them:
 * ordering[k] (k = 1..P) = pattern vertex bound k-th, negative when it is
 * reached through an in-edge of its parent; parents[v] = DFS parent of vertex
 * v, -1 for vertex 1.  Returns 0, or -1 when some vertex cannot be reached
 * from vertex 1 (the reference then reads an unset slot). */
int makeGraphOrdering(const Pattern *p, int32_t ordering[MAX_MATCH_SIZE + 1], int32_t parents[MAX_MATCH_SIZE + 1]);

/* Options for buildPlan. */
enum {
    PLAN_SYMMETRIC_DATA = 1, /* in-lists equal out-lists: every constraint uses the out-CSR */
    PLAN_BREAK_SYMMETRY = 2, /* add ordering bounds from the pattern's automorphisms (simple data graphs only) */
    PLAN_DEGREE_FILTER = 4,  /* require deg(c) >= deg_P(p) (simple data graphs only) */
    PLAN_REFERENCE_ORDER = 8 /* bind the pattern vertices in the reference's depth-first order (also GPM_PLAN_ORDER=reference);
                              * default: same order with the leaves moved to the end */
};

/* Degree statistics of the data graph for the cost model of the binding order: the distinct values of the
 * (undirected) degree, capped at sqrt(sum of degrees) so that the expected-count formula below stays
 * meaningful on heavy-tailed graphs (an arc between two vertices exists with probability at most one),
 * and how many vertices have each. */
typedef struct {
    int32_t nValues;
    double *value; /* ascending */
    double *count;
    double moments[MAX_MATCH_SIZE];   /* moments[j] = sum over the vertices of degree^j */
    double *cdf[MAX_MATCH_SIZE];      /* cdf[j][i] = sum over values[0..i] of count * value^j */
} GraphStats;

GraphStats *graphStatsNew(int32_t n, const int64_t *outOffsets, const int64_t *inOffsets);
void graphStatsFree(GraphStats *s);

/* As buildPlan, with the binding order chosen by cost when the data graph is simple and symmetric
 * (PLAN_SYMMETRIC_DATA and PLAN_DEGREE_FILTER set), stats is given and neither PLAN_REFERENCE_ORDER nor
 * GPM_PLAN_ORDER asks for another order.
 *
 * Cost of an order = expected number of partial matches the search handles one by one, summed over the
 * levels.  For the partial pattern H on the first k vertices: the product over its vertices v of
 * moments[deg_H(v)], divided by moments[1]^|E(H)| -- the expected copies of H in a random graph with the
 * data graph's degrees -- times the probability that the symmetry-breaking bounds among those vertices
 * hold.  The library orders vertices by degree, so a bound m(a) < m(b) says "a has the lower degree"; the
 * probability is estimated by drawing every vertex's degree from the degree distribution weighted by
 * degree^deg_H(v) (fixed seed).  This is what makes extending a partial match from the lower end of a bound
 * much cheaper than from the upper end on a skewed graph.  Levels counted in closed form cost nothing:
 * leaves bound last, and the last two levels when their vertices are not adjacent (independent or twin
 * candidate sets, counted per partial match of the level before); otherwise the last level alone is
 * fused into the one before it.  Candidates are all orders of the non-leaf vertices that keep every
 * prefix connected (patterns of 4 to 8 such vertices) and, when the default order ends in two non-adjacent
 * vertices, end in two non-adjacent vertices as well -- the model underestimates dense partial patterns on
 * graphs with a dense core, so it is not allowed to add a level that is handled row by row; the default order (reference numbering with leaves
 * and twins last) is kept unless the best candidate is expected to do less than half its work. */
int buildPlanForGraph(const Pattern *p, int flags, const GraphStats *stats, gpm_plan *plan, char *err, int errLen);

/* Fill *plan for pattern p.  On success plan->aut points to a malloc'd table
 * the caller frees with free() (NULL when PLAN_BREAK_SYMMETRY is off or the
 * pattern has no symmetry).  Returns 0 or -1 with a message in err. */
int buildPlan(const Pattern *p, int flags, gpm_plan *plan, char *err, int errLen);

#endif
