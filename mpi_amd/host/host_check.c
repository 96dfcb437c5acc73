/*
 * host_check -- runs the host-side code of gpm-par (input parser, threaded
 * loader, CSR cache, planner) on input files and prints a digest of what it
 * built.  It exists so that this code can run under AddressSanitizer and
 * UndefinedBehaviorSanitizer (`make sanitize`): the digest of the sanitised
 * build must equal the plain build's, and the sanitisers must stay silent.
 *
 *     host_check <inFile> ...
 */
#define _POSIX_C_SOURCE 200809L
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "graph.h"
#include "plan.h"

static uint64_t mix(uint64_t h, uint64_t x) {
    h ^= x + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    return h * 0xbf58476d1ce4e5b9ull;
}

static uint64_t digestPlan(const gpm_plan *p) {
    uint64_t h = (uint64_t)p->n_levels;
    for (int k = 0; k < p->n_levels; ++k) {
        h = mix(h, (uint64_t)p->order[k]);
        h = mix(h, (uint64_t)(p->src_level[k] + 1) * 4 + (uint64_t)p->src_dir[k]);
        h = mix(h, (uint64_t)p->label[k] + 7);
        h = mix(h, (uint64_t)p->min_out_deg[k] * 16 + (uint64_t)p->min_in_deg[k]);
        for (int j = 0; j < p->n_cons[k]; ++j) h = mix(h, (uint64_t)p->cons_level[k][j] * 2 + (uint64_t)p->cons_dir[k][j]);
        for (int j = 0; j < p->n_lower[k]; ++j) h = mix(h, 100u + (uint64_t)p->lower_level[k][j]);
        for (int j = 0; j < p->n_upper[k]; ++j) h = mix(h, 200u + (uint64_t)p->upper_level[k][j]);
    }
    h = mix(h, p->n_aut);
    if (p->aut)
        for (uint64_t i = 0; i < p->n_aut * (uint64_t)p->n_levels; ++i) h = mix(h, (uint64_t)(uint8_t)p->aut[i]);
    return h;
}

int main(int argc, char *argv[]) {
    for (int a = 1; a < argc; ++a) {
        char err[256];
        Graph *g = NULL;
        Pattern *p = NULL;
        if (readInput(argv[a], &g, &p, err, sizeof err) != 0) {
            printf("%s: error: %s\n", argv[a], err);
            continue;
        }
        uint64_t h = mix((uint64_t)g->nodesCount, (uint64_t)g->edgesCount);
        for (int32_t v = 0; v <= g->nodesCount; ++v) h = mix(mix(h, (uint64_t)g->outOffsets[v]), (uint64_t)g->inOffsets[v]);
        for (int64_t e = 0; e < g->edgesCount; ++e) h = mix(mix(h, (uint64_t)g->outEdges[e]), (uint64_t)g->inEdges[e]);
        if (g->labels)
            for (int32_t v = 0; v < g->nodesCount; ++v) h = mix(h, g->labels[v]);
        printf("%s: n %d arcs %" PRId64 " symmetric %d simple %d labels %d graph %016" PRIx64, argv[a], g->nodesCount,
               g->edgesCount, g->symmetric, g->simple, g->labels != NULL, h);
        int32_t ordering[MAX_MATCH_SIZE + 1], parents[MAX_MATCH_SIZE + 1];
        int connected = makeGraphOrdering(p, ordering, parents) == 0;
        printf(" pattern %d connected %d", p->nodesCount, connected);
        const int sym = g->symmetric && g->simple ? PLAN_SYMMETRIC_DATA : 0;
        const int flagSets[4] = {sym, sym | PLAN_DEGREE_FILTER | PLAN_BREAK_SYMMETRY, sym | PLAN_REFERENCE_ORDER,
                                 PLAN_BREAK_SYMMETRY | PLAN_REFERENCE_ORDER};
        for (int f = 0; f < 4; ++f) {
            gpm_plan plan;
            if (buildPlan(p, flagSets[f], &plan, err, sizeof err) != 0) {
                printf(" plan%d error", f);
                continue;
            }
            printf(" plan%d %016" PRIx64, f, digestPlan(&plan));
            free((void *)plan.aut);
        }
        printf("\n");
        freeGraph(g);
        freePattern(p);
    }
    return 0;
}
