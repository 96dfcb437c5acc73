/*
 * ref_isend_fix.c -- harness-side workaround, linked next to the untouched
 * reference sources when building oracle/_ref/gpm-par-ref-k.
 *
 * TEST / BASELINE INFRASTRUCTURE ONLY.
 *
 * With three or more ranks the reference posts one 2-int MPI_Isend per
 * remote edge from a scratch array, drops the request handle at once and
 * frees the array while sends may still be queued (reference:
 * src/gpm-par.c:536-537 and :551).  The queued sends then read freed
 * memory and the program crashes.  Defining MPI_Isend here shadows the MPI
 * library's weak symbol: sends with that tag (110, MPI_NODE_IN_NEIGH,
 * src/gpm-par.c:19) are issued from a private copy of the payload that is
 * never freed; every other send is forwarded unchanged through PMPI_Isend.
 * The algorithm, message order and results are untouched.
 */
#include <mpi.h>
#include <stdlib.h>
#include <string.h>

int MPI_Isend(const void *buf, int count, MPI_Datatype type, int dest, int tag,
              MPI_Comm comm, MPI_Request *req) {
    if (tag == 110 && count > 0 && buf) {
        int sz = 0;
        MPI_Type_size(type, &sz);
        void *keep = malloc((size_t)count * (size_t)sz); /* leaked on purpose */
        if (keep) {
            memcpy(keep, buf, (size_t)count * (size_t)sz);
            buf = keep;
        }
    }
    return PMPI_Isend(buf, count, type, dest, tag, comm, req);
}
