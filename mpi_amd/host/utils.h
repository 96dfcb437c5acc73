/* Small host helpers (counterpart of the reference's src/utils.h). */
#ifndef GPM_HOST_UTILS_H_
#define GPM_HOST_UTILS_H_

#include <stddef.h>
#include <stdint.h>

/* Allocation that ends the program the way the reference does on failure:
 * "ERROR: malloc returned NULL" on stderr, exit code 1 (reference:
 * safeMalloc, src/utils.c:20-28).  Sizes are size_t, so lists beyond 2 GiB
 * work. */
void *safeMalloc(size_t size);
void *safeCalloc(size_t count, size_t size);
void *safeRealloc(void *p, size_t size);

/* 1 when [s, e) holds only whitespace (reference: isLineEmpty,
 * src/utils.c:9-18). */
int isLineEmpty(const char *s, const char *e);

/* Position of val in the ascending array tab[0..len), or -1 (0-based
 * counterpart of the reference's 1-based findIndex, src/utils.c:30-44, which
 * also misreads an empty table). */
int64_t findIndex(const int32_t *tab, int64_t len, int32_t val);

/* Membership in an ascending array by binary search (the reference scans
 * unsorted lists linearly: arrayContains, src/utils.c:46-53). */
int arrayContains(const int32_t *tab, int64_t len, int32_t val);

/* Ascending in-place sort of 32-bit ids, duplicates kept. */
void sortIds(int32_t *a, int64_t n);

double wallSeconds(void);

/* Take the next root chunk from a cursor shared by every GPU worker of the
 * node (threads of one process, or processes mapping the same memory):
 * returns the chunk index, atomically incremented.  This replaces static
 * vertex ownership (reference: assignNodesToProcesses, src/gpm-par.c:300-340)
 * with work stealing. */
uint64_t claimChunk(volatile uint64_t *cursor);

#endif
