/*
 * gpm-par -- find every occurrence of a small pattern graph in a large data
 * graph, on MI355X GPUs.
 *
 * Drop-in for the reference program's command line, input format, output
 * format and stdout (reference: src/gpm-par.c:49-69, :555-648):
 *
 *     gpm-par.exe <inFile> <outFile>
 *
 * One plain process drives 1..8 GPUs; it needs no mpirun.  When it is started
 * under mpirun anyway, every rank but 0 exits at once so that the launch line
 * the reference needs still produces one correct output file.
 *
 * Knobs are environment variables so that argc == 3 stays the contract:
 *   GPM_NGPUS       GPUs to use (default: all visible, at most 8)
 *   GPM_COUNT_ONLY  1 = count matches, write nothing; the count goes to stderr
 *   GPM_SYMBREAK    0 = enumerate every automorphic image on the GPU instead
 *                   of expanding canonical matches (default 1)
 *   GPM_STATS       1 = print search statistics to stderr
 *   GPM_HOST_FORMAT 1 = format the output lines on the host instead of the GPU
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/gpm_hip.h"
#include "graph.h"
#include "plan.h"
#include "utils.h"

static int envInt(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

/* Rank of this process when launched by an MPI starter, else 0. */
static int launcherRank(void) {
    static const char *vars[] = {"PMI_RANK", "PMIX_RANK", "OMPI_COMM_WORLD_RANK", "MV2_COMM_WORLD_RANK"};
    for (size_t i = 0; i < sizeof vars / sizeof vars[0]; ++i) {
        const char *v = getenv(vars[i]);
        if (v && *v) return atoi(v);
    }
    return 0;
}

/* reference: checkArgs, src/gpm-par.c:49-69 (same messages and exit code;
 * the reference goes on to fclose(NULL) when the input is missing). */
static void checkArgs(int argc, char *argv[], FILE **outFile) {
    if (argc != 3) {
        fprintf(stderr, "ERROR: files not specified\n");
        exit(1);
    }
    FILE *in = fopen(argv[1], "r");
    if (in == NULL) {
        fprintf(stderr, "ERROR: file not found %s\n", argv[1]);
        exit(1);
    }
    fclose(in);
    *outFile = fopen(argv[2], "w+");
    if (*outFile == NULL) {
        fprintf(stderr, "ERROR: file not found %s\n", argv[2]);
        exit(1);
    }
}

static void die(const char *what, const char *detail) {
    fprintf(stderr, "ERROR: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
    exit(1);
}

typedef struct {
    FILE *out;
    char *buf;
    size_t len, cap;
} Writer;

/* One line per match, ids separated by single spaces (reference: writeMatch,
 * src/gpm-par.c:71-77); formatted by hand into a large buffer because a
 * printf per match would dominate the run time. */
static int writeMatches(void *user, const int32_t *rows, uint64_t nrows, int32_t ncols) {
    Writer *w = user;
    for (uint64_t r = 0; r < nrows; ++r) {
        if (w->len + (size_t)ncols * 12 + 2 > w->cap) {
            if (fwrite(w->buf, 1, w->len, w->out) != w->len) return 1;
            w->len = 0;
        }
        const int32_t *m = rows + r * (uint64_t)ncols;
        for (int32_t i = 0; i < ncols; ++i) {
            char tmp[12];
            int n = 0;
            uint32_t v = (uint32_t)m[i];
            do {
                tmp[n++] = (char)('0' + v % 10);
                v /= 10;
            } while (v);
            if (i) w->buf[w->len++] = ' ';
            while (n) w->buf[w->len++] = tmp[--n];
        }
        w->buf[w->len++] = '\n';
    }
    return 0;
}

/* Lines formatted on the GPU arrive as blocks of text. */
static int writeText(void *user, const char *text, size_t bytes, uint64_t nlines) {
    Writer *w = user;
    (void)nlines;
    return fwrite(text, 1, bytes, w->out) == bytes ? 0 : 1;
}

int main(int argc, char *argv[]) {
    if (launcherRank() != 0) return 0;

    double startDistTime = wallSeconds();
    FILE *outFile;
    checkArgs(argc, argv, &outFile);

    char err[256];
    Graph *graph;
    Pattern *pattern;
    if (readInput(argv[1], &graph, &pattern, err, sizeof err) != 0) die("bad input", err);
    double loadedTime = wallSeconds();

    int ngpus = gpm_hip_device_count();
    if (ngpus <= 0) die("no GPU found", gpm_hip_last_error());
    if (ngpus > 8) ngpus = 8;
    int want = envInt("GPM_NGPUS", ngpus);
    if (want >= 1 && want < ngpus) ngpus = want;

    gpm_ctx *ctx;
    if (gpm_hip_create(ngpus, NULL, &ctx) != 0) die("GPU set-up failed", gpm_hip_last_error());
    /* The in-CSR may be left out only when in-lists equal out-lists entry for entry.  `symmetric` says
     * they are equal as sets; with duplicate arcs the multiplicities can differ, and a level reached
     * over an in-edge walks the in-list with multiplicity (reference: src/gpm-par.c:205-213). */
    int sameLists = graph->symmetric && graph->simple;
    if (gpm_hip_upload_graph(ctx, graph->nodesCount, graph->outOffsets, graph->outEdges,
                             sameLists ? NULL : graph->inOffsets, sameLists ? NULL : graph->inEdges,
                             graph->labels) != 0)
        die("graph upload failed", gpm_hip_last_error());
    double uploadedTime = wallSeconds();

    int flags = 0;
    if (sameLists) flags |= PLAN_SYMMETRIC_DATA;
    /* With duplicate arcs the number of times a match is printed depends on which pattern arcs are walked
     * (with multiplicity) and which are membership tests: keep the reference's tree. */
    if (!graph->simple) flags |= PLAN_REFERENCE_ORDER;
    if (graph->simple) flags |= PLAN_DEGREE_FILTER;
    if (graph->simple && envInt("GPM_SYMBREAK", 1)) flags |= PLAN_BREAK_SYMMETRY;
    gpm_plan plan;
    /* the binding order is chosen by cost from the data graph's degree distribution (simple symmetric graphs) */
    GraphStats *stats = sameLists ? graphStatsNew(graph->nodesCount, graph->outOffsets, NULL) : NULL;
    if (buildPlanForGraph(pattern, flags, stats, &plan, err, sizeof err) != 0) die("bad pattern", err);
    graphStatsFree(stats);
    if (gpm_hip_set_plan(ctx, &plan) != 0) die("plan rejected", gpm_hip_last_error());
    double plannedTime = wallSeconds();
    printf("Distribution time[s]: %d\n", (int)(wallSeconds() - startDistTime));
    fflush(stdout);

    double startComputationTime = wallSeconds();
    int countOnly = envInt("GPM_COUNT_ONLY", 0);
    Writer w = {outFile, safeMalloc(1u << 22), 0, 1u << 22};
    uint64_t total = 0;
    int rc;
    if (!countOnly && !envInt("GPM_HOST_FORMAT", 0)) rc = gpm_hip_search_text(ctx, writeText, &w, &total);
    else rc = gpm_hip_search(ctx, countOnly, countOnly ? NULL : writeMatches, &w, &total);
    if (rc != 0) die("search failed", gpm_hip_last_error());
    if (w.len && fwrite(w.buf, 1, w.len, outFile) != w.len) die("write failed", argv[2]);
    fclose(outFile);
    printf("Computations time[s]: %d\n", (int)(wallSeconds() - startComputationTime));

    if (countOnly) fprintf(stderr, "matches: %llu\n", (unsigned long long)total);
    if (envInt("GPM_STATS", 0)) {
        gpm_stats st;
        gpm_hip_stats(ctx, &st);
        /* what "Distribution time" is made of: reading the input, GPU set-up + upload + renumbering, planning */
        fprintf(stderr, "load_s %.3f upload_s %.3f plan_s %.3f\n", loadedTime - startDistTime,
                uploadedTime - loadedTime, plannedTime - uploadedTime);
        fprintf(stderr, "matches %llu canonical %llu alg_bytes %llu search_ms %.3f tail_ms %.3f extend_ms %.3f\n",
                (unsigned long long)st.matches, (unsigned long long)st.canonical, (unsigned long long)st.alg_bytes,
                st.ms_total, st.ms_tail_kernel, st.ms_extend_kernel);
    }
    gpm_hip_destroy(ctx);
    free((void *)plan.aut);
    free(w.buf);
    freeGraph(graph);
    freePattern(pattern);
    return 0;
}
