/* Multi-threaded fast path of the input parser (see loader.c). */
#ifndef GPM_HOST_LOADER_H_
#define GPM_HOST_LOADER_H_

#include <stdint.h>

#include "graph.h"

/* Threads the loader may use: GPM_LOAD_THREADS, default the online CPUs up to 16. */
int loaderThreads(void);

/* Parse the data-graph section that starts at `begin` (just after the vertex
 * count n and the whitespace that follows it) with `threads` threads and
 * build the CSR graph.  On success returns the graph, sets *next to the
 * first byte after the blank line that ends the section and *lines to the
 * number of lines consumed.  Returns NULL, touching nothing, when the input
 * is not plain well-formed text of the format; the caller then runs the
 * single-threaded parser, which owns the error messages. */
Graph *loadDataSectionParallel(const char *begin, const char *end, int64_t n, int threads, const char **next,
                               int64_t *lines);

#endif
