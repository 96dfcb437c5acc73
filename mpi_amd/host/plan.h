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

/* Fill *plan for pattern p.  On success plan->aut points to a malloc'd table
 * the caller frees with free() (NULL when PLAN_BREAK_SYMMETRY is off or the
 * pattern has no symmetry).  Returns 0 or -1 with a message in err. */
int buildPlan(const Pattern *p, int flags, gpm_plan *plan, char *err, int errLen);

#endif
