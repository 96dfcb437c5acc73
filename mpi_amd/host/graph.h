/*
 * Host-side graph model and input parser (counterpart of the reference's
 * src/graph.h + the two file passes of src/gpm-par.c).  The data graph is
 * kept as CSR in both directions, ready to be copied into HBM; the pattern
 * stays a small jagged graph in listing order because the matching plan
 * depends on that order.
 */
#ifndef GPM_HOST_GRAPH_H_
#define GPM_HOST_GRAPH_H_

#include <stdint.h>

#define MAX_MATCH_SIZE 10 /* reference: src/gpm-par.c:13 */

/* Data graph (reference: Graph, src/graph.h:6-15).  Ids are 0-based here
 * (file id - 1).  Lists are sorted ascending; duplicate arcs are kept because
 * a neighbour listed twice is tried twice (reference: src/gpm-par.c:212-213). */
typedef struct {
    int32_t nodesCount;
    int64_t edgesCount;
    int64_t *outOffsets; /* nodesCount + 1 */
    int32_t *outEdges;
    int64_t *inOffsets;  /* transpose */
    int32_t *inEdges;
    uint8_t *labels;     /* nodesCount labels, or NULL */
    int symmetric;       /* every in-list equals the out-list as a set */
    int simple;          /* no duplicate arcs, no self loops */
} Graph;

/* Pattern graph: at most MAX_MATCH_SIZE vertices, lists in listing order
 * (out) and ascending source order (in), duplicates and self loops kept as
 * listed (reference: readGraph, src/graph.c:38-86). */
typedef struct {
    int32_t nodesCount;
    int32_t *outEdges[MAX_MATCH_SIZE];
    int32_t outDegrees[MAX_MATCH_SIZE];
    int32_t *inEdges[MAX_MATCH_SIZE];
    int32_t inDegrees[MAX_MATCH_SIZE];
    int32_t labels[MAX_MATCH_SIZE]; /* -1 = unlabelled */
    int hasLabels;
} Pattern;

/* Parse a whole input file: data graph, blank line, pattern, optional
 * "labels" trailer.  Returns 0, or -1 with a message in err (errLen bytes). */
int readInput(const char *path, Graph **data, Pattern **pattern, char *err, int errLen);
int readInputBuffer(const char *buf, int64_t len, Graph **data, Pattern **pattern, char *err, int errLen);

/* Write the data-graph section of the input format (vertex count, one block
 * per vertex with out-arcs, the blank line that ends the section) from a CSR
 * with 0-based ids; the caller appends the pattern section.  Used by the
 * benchmark generators.  Returns 0, or -1 on an I/O error. */
int writeDataSection(const char *path, int32_t n, const int64_t *outOffsets, const int32_t *outEdges);

/* Build a data graph from 0-based arcs (any order, duplicates allowed). */
Graph *graphFromEdges(int32_t n, const int32_t *src, const int32_t *dst, int64_t m);
/* Build a pattern from 0-based arcs in listing order. */
Pattern *patternFromEdges(int32_t n, const int32_t *src, const int32_t *dst, int32_t m);

/* Attach one label (0..255) per pattern vertex, or NULL to remove them. */
void patternSetLabels(Pattern *p, const int32_t *labels);

void freeGraph(Graph *g);
void freePattern(Pattern *p);
void printGraph(const Graph *g);

#endif
