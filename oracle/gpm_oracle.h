/*
 * gpm_oracle.h -- CPU oracle for the gpm-par match search.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a single-threaded restatement of the
 * reference algorithm (reference: src/gpm-par.c, src/graph.c, src/utils.c).
 * It exists to check the HIP path; nothing on the product path links,
 * imports or executes it.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it.
 *
 * Parity pin: the oracle's output file is compared byte-for-byte with the
 * compiled reference run as `mpirun -np 2` (one worker => deterministic
 * order) on every fixture under tests/golden/ (see tests/test_oracle.py and
 * tests/golden/make_golden.py).
 */
#ifndef GPM_ORACLE_H_
#define GPM_ORACLE_H_

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_PATTERN 10 /* reference: MAX_MATCH_SIZE, src/gpm-par.c:13 */

/* Jagged directed graph, 1-based vertex ids, as the reference keeps it
 * (reference: Graph, src/graph.h:6-15).  Lists keep file order and
 * duplicates. */
typedef struct {
    int n;        /* vertex count, ids 1..n */
    int **out;    /* out[v][0..outdeg[v]) */
    int *outdeg;
    int **in;     /* in[v][0..indeg[v]), ascending source id (transpose order) */
    int *indeg;
    int *ordering; /* pattern only: signed matching order, 1..n */
    int *parents;  /* pattern only: DFS parent of each vertex, -1 for vertex 1 */
} OGraph;

typedef struct {
    long long matches;        /* embeddings found */
    long long extended[ORACLE_MAX_PATTERN + 2]; /* extended[k] = partial matches of length k that were expanded */
    long long alg_bytes;      /* merge-model algorithmic bytes (see oracle_search) */
} OStats;

/* Build a graph from an edge list given in listing order (1-based ids).
 * Out-lists keep the order edges appear in; in-lists are the transpose in
 * ascending source order (reference: src/graph.c:68-83). */
OGraph *oracle_graph_from_edges(int n, const int *src, const int *dst, long long m);
void oracle_graph_free(OGraph *g);

/* Compute the matching plan of a pattern (reference: dfs/makeGraphOrdering,
 * src/graph.c:7-36).  Returns 0, or -1 if some vertex is unreachable from
 * vertex 1 (the reference reads garbage in that case). */
int oracle_make_plan(OGraph *pattern);

/* Parse one input file: data graph, blank line, pattern
 * (reference: preprocessGraph src/gpm-par.c:267-286, readGraph
 * src/graph.c:38-86).  Optional trailer after the pattern's blank line:
 * "labels" line, then n data labels and P pattern labels, one per line
 * (an extension the reference ignores).  labels_out may be NULL. */
int oracle_parse_file(const char *path, OGraph **data, OGraph **pattern,
                      int **data_labels, int **pattern_labels);

/* Enumerate all matches in the reference's one-worker order: roots
 * ascending, depth-first, candidates in adjacency order (reference:
 * findMatches/exploreMatch/checkNodeMatches, src/gpm-par.c:88-112,179-252).
 * Each match is a row of P ints in pattern-vertex order (reference:
 * writeMatch, src/gpm-par.c:71-77).
 *   rows/cap : optional output buffer of cap rows (rows beyond cap are
 *              counted but not stored); rows may be NULL.
 *   out      : optional FILE* receiving one line per match.
 *   fast     : 0 = linear membership scans like the reference
 *              (utils.c:46-53); 1 = binary search on sorted copies (same
 *              results and order, used for larger test inputs); 2 = as 1,
 *              and with labels a vertex's label is tested when the vertex is
 *              bound, which drops whole subtrees the post-filter would have
 *              enumerated and thrown away (same matches, same order; the
 *              statistics then describe the pruned search).
 *   data_labels/pattern_labels: NULL, or label filter (a match is kept iff
 *              label_G(m(i)) == label_P(i) for every i); applied as a
 *              post-filter on complete matches.
 *   root_lo/root_hi: restrict roots to [root_lo, root_hi] (1-based,
 *              inclusive); pass 1 and n for the full search.
 * Returns the number of matches, or -1 on error. */
long long oracle_search(const OGraph *data, const OGraph *pattern, int fast,
                        const int *data_labels, const int *pattern_labels,
                        int root_lo, int root_hi,
                        int *rows, long long cap, FILE *out, OStats *stats);

/* Whole-program equivalent of `gpm-par.exe in out` (without the timing
 * lines).  Returns match count or -1. */
long long oracle_run_file(const char *in_path, const char *out_path, int fast);

#ifdef __cplusplus
}
#endif
#endif
