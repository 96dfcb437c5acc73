/*
 * gpm_hip.h -- C ABI of the MI355X match-search library (libgpm_hip.so).
 *
 * The reference has no plugin or FFI interface; its search is one C call,
 * findMatches(graph, pattern, nodeProcMap, rank) (reference:
 * src/gpm-par.c:245, called at :628), whose results leave the worker as one
 * message of P+1 ints per match (reference: src/gpm-par.c:181) that rank 0
 * turns into one output line (reference: src/gpm-par.c:608-613).  This
 * header is that seam as a C ABI: plain pointers and sizes, no C++ or
 * framework types.  A C host (mpi_amd/host/gpm-par.c) calls it exactly where
 * the reference calls findMatches.
 *
 * Conventions
 *   - every function returns 0 on success and a non-zero GPM_E* code on
 *     error; gpm_hip_last_error() returns a message for the calling thread;
 *   - the caller owns every host pointer it passes; device memory belongs to
 *     the context and is released by gpm_hip_destroy();
 *   - vertex ids are 0-based at this boundary (the text format's ids minus 1);
 *   - the library never calls back into host code from device threads; the
 *     row sink of gpm_hip_search() runs on the calling host thread;
 *   - one context may drive several GPUs of one node; calls on one context
 *     must not overlap.  In output mode the match lists of all GPUs are
 *     gathered on GPU 0 in rounds -- RCCL all-gather of the row counts and of
 *     the lists padded to the longest, or peer copies when the context has no
 *     communicators -- and GPU 0's copy goes to the sink (replaces the message
 *     per match to rank 0, reference: src/gpm-par.c:181, :608-621).
 *     GPM_GATHER=host makes every GPU hand its own rows to the host instead.
 */
#ifndef GPM_HIP_H_
#define GPM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPM_MAX_PATTERN 10 /* reference: MAX_MATCH_SIZE, src/gpm-par.c:13 */
#define GPM_MAX_CONS 9     /* a level has at most P-1 matched neighbours */

enum {
    GPM_OK = 0,
    GPM_EINVAL = 1,   /* bad argument (shape, range, NULL) */
    GPM_ENODEV = 2,   /* no usable GPU / HIP runtime error at set-up */
    GPM_ENOMEM = 3,   /* host or device allocation failed */
    GPM_ERUNTIME = 4, /* HIP error during a call */
    GPM_ESTATE = 5,   /* call order violated (e.g. search before upload) */
    GPM_ESINK = 6     /* the row sink asked to stop */
};

typedef struct gpm_ctx gpm_ctx;

/*
 * Matching plan: the order in which pattern vertices are bound and, for each
 * level, where candidates come from and what they must satisfy.  It restates
 * what the reference derives on the fly from Graph.ordering / Graph.parents
 * (reference: src/graph.c:7-36, used at src/gpm-par.c:186-206) and from the
 * pattern rows in checkNodeMatches (reference: src/gpm-par.c:88-112).
 *
 * Level 0 binds order[0] to a root vertex.  Level k >= 1 binds order[k]:
 *   candidates   = N_dir(m[src_level[k]]) walked WITH multiplicity, where
 *                  dir = src_dir[k] (0 = out-list, 1 = in-list);
 *   accepted iff   c is not already in the partial match, and
 *                  c is in N_{cons_dir[k][j]}(m[cons_level[k][j]]) for every j, and
 *                  c >  m[lower_level[k][j]] for every j  (symmetry breaking), and
 *                  c <  m[upper_level[k][j]] for every j  (symmetry breaking), and
 *                  label/degree filters hold (see below).
 * "m[i]" is the data vertex bound at level i.  The order relation of the
 * symmetry-breaking bounds is the library's internal vertex order (any strict
 * total order gives the same result set); such bounds are only legal on a data
 * graph without duplicate arcs, and the number of automorphic images each
 * reported match stands for is n_aut (1 when no bounds are used).
 */
typedef struct {
    int32_t n_levels;                 /* P, 1..GPM_MAX_PATTERN */
    int32_t order[GPM_MAX_PATTERN];   /* pattern vertex (0-based) bound at level k */
    int32_t src_level[GPM_MAX_PATTERN];
    int32_t src_dir[GPM_MAX_PATTERN];
    int32_t n_cons[GPM_MAX_PATTERN];
    int32_t cons_level[GPM_MAX_PATTERN][GPM_MAX_CONS];
    int32_t cons_dir[GPM_MAX_PATTERN][GPM_MAX_CONS];
    int32_t n_lower[GPM_MAX_PATTERN];
    int32_t lower_level[GPM_MAX_PATTERN][GPM_MAX_CONS];
    int32_t n_upper[GPM_MAX_PATTERN];
    int32_t upper_level[GPM_MAX_PATTERN][GPM_MAX_CONS];
    /* filters per level; label < 0 means "any"; degrees count distinct
     * non-self neighbours the data vertex must at least have */
    int32_t label[GPM_MAX_PATTERN];
    int32_t min_out_deg[GPM_MAX_PATTERN];
    int32_t min_in_deg[GPM_MAX_PATTERN];
    /* automorphisms used to expand each reported match into all its images:
     * aut[i*P + v] = image of pattern vertex v under automorphism i; the
     * identity comes first.  n_aut == 1 and aut == NULL when unused. */
    uint64_t n_aut;
    const int8_t *aut;
} gpm_plan;

typedef struct {
    int32_t n;           /* vertices */
    int64_t m_out;       /* arcs */
    int32_t symmetric;   /* in-lists equal out-lists as sets */
    int32_t simple;      /* no duplicate arcs and no self loops */
    int32_t max_out_deg, max_in_deg;
    int32_t has_labels;
    int32_t hub_rows;    /* rows of the dense bitmap over the highest-degree vertices on GPU 0 (0: none was
                          * allocated; every kernel then tests membership in the lists) */
} gpm_graph_info;

/* Kernel families the search is made of; indices of the kernel_* arrays of gpm_stats. */
#define GPM_KERNEL_KINDS 8
enum {
    GPM_KIND_TAIL = 0,      /* fused last two levels: list intersection through LDS, closed-form leaves */
    GPM_KIND_EXTEND = 1,    /* level-by-level extension with ballot / prefix-sum compaction */
    GPM_KIND_CLIQUE = 2,    /* cliques, small universes: bitmap rows built and counted in LDS */
    GPM_KIND_CLQ_BUILD = 3, /* cliques, large universes: bitmap rows built into HBM */
    GPM_KIND_CLQ_COUNT = 4, /* cliques, large universes: rows counted from LDS tiles */
    GPM_KIND_CYCLE4 = 5,    /* 4-cycles: two-path counters */
    GPM_KIND_OCTA = 6       /* octahedra (K6 minus a perfect matching): 4-cycles among common higher neighbours */
};

typedef struct {
    uint64_t matches;                      /* embeddings (all automorphic images) */
    uint64_t canonical;                    /* matches before multiplying by n_aut */
    uint64_t rows_in[GPM_MAX_PATTERN];     /* partial matches of length k+1 that were extended */
    uint64_t alg_bytes;                    /* merge-model bytes of the work done: per extended partial
                                            * match, its ids plus one pass over every operand list */
    uint64_t tail_alg_bytes;               /* part of alg_bytes done by the fused intersection kernel */
    double   ms_total;                     /* wall time of the last search */
    double   ms_tail_kernel;               /* summed device time of the intersection kernel */
    double   ms_extend_kernel;             /* summed device time of the level-extension kernel */
    uint64_t tail_launches, extend_launches;
    uint64_t chunks[8];                    /* root chunks taken by each GPU of the context */
    /* per kernel family: summed device time, launches, merge-model bytes of the work done */
    double   kernel_ms[GPM_KERNEL_KINDS];
    uint64_t kernel_launches[GPM_KERNEL_KINDS];
    uint64_t kernel_alg_bytes[GPM_KERNEL_KINDS];
    /* output mode on several GPUs: gather rounds run (meetings of all workers) and canonical rows that
     * reached the writer through GPU 0 (RCCL all-gather, or peer copies without communicators) */
    uint64_t gather_rounds, gathered_rows;
} gpm_stats;

/* Row sink: called on the calling host thread with `nrows` complete matches,
 * `ncols` = P ints each, in pattern-vertex order, 1-based ids (the numbers the
 * reference prints, reference: writeMatch src/gpm-par.c:71-77).  Return 0 to
 * continue, non-zero to stop the search (gpm_hip_search then returns
 * GPM_ESINK). */
typedef int (*gpm_row_sink)(void *user, const int32_t *rows, uint64_t nrows, int32_t ncols);

/* Text sink: called on the calling host thread with `bytes` bytes holding
 * `nlines` whole output lines, "m(1) m(2) ... m(P)\n" with 1-based ids -- the
 * bytes the reference's writeMatch prints (reference: src/gpm-par.c:71-77) --
 * formatted on the GPU.  The buffer is only valid during the call.  Return 0
 * to continue, non-zero to stop the search. */
typedef int (*gpm_text_sink)(void *user, const char *text, size_t bytes, uint64_t nlines);

const char *gpm_hip_last_error(void);

/* Number of GPUs the runtime sees (0 when there is none). */
int gpm_hip_device_count(void);

/* Create a context on `ngpus` devices.  devices == NULL selects 0..ngpus-1.
 * Replaces MPI_Init / Comm_size / Comm_rank (reference: src/gpm-par.c:636-639). */
int gpm_hip_create(int ngpus, const int *devices, gpm_ctx **out);
void gpm_hip_destroy(gpm_ctx *ctx);

/* Copy the data graph to every GPU of the context as CSR (replaces the
 * scatter of the graph over ranks, reference: distribiuteGraph / gatherGraph /
 * transpose, src/gpm-par.c:394-553).  Lists must be sorted ascending;
 * duplicates are kept.  in_off/in_adj may be NULL when every in-list equals
 * the out-list of the same vertex entry for entry, duplicates included (the
 * in-CSR then aliases the out-CSR); a graph that is symmetric only as a set of
 * arcs but lists some arc more often than its reverse needs both.  labels may
 * be NULL. */
int gpm_hip_upload_graph(gpm_ctx *ctx, int32_t n,
                         const int64_t *out_off, const int32_t *out_adj,
                         const int64_t *in_off, const int32_t *in_adj,
                         const uint8_t *labels);
int gpm_hip_graph_info(const gpm_ctx *ctx, gpm_graph_info *info);

/* Install the plan (replaces the pattern broadcast, reference:
 * distribiutePattern / gatherPattern, src/gpm-par.c:342-392). */
int gpm_hip_set_plan(gpm_ctx *ctx, const gpm_plan *plan);

/* Share the root-chunk cursor with other processes on this node (one context
 * per process, one GPU each): `cursor` points to a zero-initialised 64-bit
 * word in memory all of them map.  NULL (default) = private cursor.  Together
 * with gpm_hip_set_roots this replaces vertex ownership (reference:
 * assignNodesToProcesses, src/gpm-par.c:300-340). */
int gpm_hip_set_cursor(gpm_ctx *ctx, volatile uint64_t *cursor);

/* How many chunks to cut the root range into (0 = default, 32 per GPU of the
 * context).  Chunks hold roots of similar estimated cost; processes that share
 * a cursor must pass the same value (e.g. 32 x number of processes) so that
 * they build the same chunk table. */
int gpm_hip_set_chunks(gpm_ctx *ctx, int64_t n_chunks);

/* Merge-model byte accounting of the clique kernels (gpm_stats.kernel_alg_bytes
 * of GPM_KIND_CLIQUE / CLQ_BUILD / CLQ_COUNT) on or off; default on, or the
 * value of GPM_ACCOUNTING.  The sums cost an add per adjacent pair inside the
 * kernels and two list-length lookups per universe entry, so a timed run
 * switches them off and takes the bytes -- a property of the work, not of the
 * run -- from a search with accounting on.  Results are the same either way;
 * every other kernel family always counts its bytes. */
int gpm_hip_set_accounting(gpm_ctx *ctx, int on);

/* Restrict roots to [lo, hi) of the library's root order (0 <= lo <= hi <= n);
 * default is all n.  Used for sharding tests. */
int gpm_hip_set_roots(gpm_ctx *ctx, int64_t lo, int64_t hi);

/* Run the search (replaces findMatches on the workers and the receive loop on
 * rank 0, reference: src/gpm-par.c:245-262 and :602-626).  count_only != 0:
 * only count; sink may be NULL.  Otherwise every match is delivered to `sink`
 * exactly once, in unspecified order.  *total receives this context's number
 * of embeddings. */
int gpm_hip_search(gpm_ctx *ctx, int count_only, gpm_row_sink sink, void *user, uint64_t *total);

/* Output mode with the lines formatted on the GPU (replaces the per-match
 * message and the fprintf per number on rank 0, reference: src/gpm-par.c:181,
 * :608-613, :71-77): every match reaches `sink` exactly once as one text
 * line, in blocks of whole lines, in unspecified order. */
int gpm_hip_search_text(gpm_ctx *ctx, gpm_text_sink sink, void *user, uint64_t *total);

/* Count-only search that allocates nothing.  It has no counterpart in the
 * reference; it is for callers that cannot allocate device memory while they
 * call: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same result as gpm_hip_search(ctx, 1, NULL, NULL, total)
 * on a single-GPU context whose graph and plan are installed (those calls do
 * allocate).  All working memory -- frontiers of 65536 rows per level,
 * candidate ranges, prefix sums, counters -- lies in one fixed 32 MiB array
 * that each GPU holds from library load until the process ends; the call makes
 * no hipMalloc / hipFree / hipHostMalloc and creates no events (so
 * gpm_hip_stats reports no kernel times after it).  It runs the level-by-level
 * and fused-intersection kernels only.  Limits: one GPU per context; longest
 * adjacency list (either direction) at most 65472 entries, else GPM_EINVAL
 * with a message; a context that shares its chunk cursor with other processes
 * (gpm_hip_set_cursor) is rejected with GPM_ESTATE, because this call walks
 * every root of the range itself and the processes' counts would no longer add
 * up to the whole.  Every call on one GPU shares that GPU's array: calls that
 * overlap on one GPU (from any context or thread) race. */
int gpm_hip_count_noalloc(gpm_ctx *ctx, uint64_t *total);

int gpm_hip_stats(const gpm_ctx *ctx, gpm_stats *stats);

/* Name of kernel family `kind` (GPM_KIND_*), "" when out of range. */
const char *gpm_hip_kernel_name(int kind);

#ifdef __cplusplus
}
#endif
#endif
