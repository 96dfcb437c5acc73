// Device-side views and helpers shared by the gpm kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GPM_WAVE 64
#define GPM_MAXP 10
#define GPM_MAXC 9

// CSR of the data graph as it sits in HBM.  Vertices are renumbered by
// ascending total degree, so "id order" is "degree order": the part of a
// sorted list above its owner (mid..end) holds only neighbours of equal or
// higher degree, which keeps the lists walked under ordering bounds short.
struct GraphView {
    int32_t n;
    const int64_t *off[2];  // [0] out-CSR, [1] in-CSR (aliases [0] on a symmetric graph)
    const int32_t *adj[2];
    const int64_t *mid[2];  // first position in v's list with id > v
    const uint8_t *label;   // by internal id, or nullptr
    // Hub matrix: the arcs among the hub_rows highest-degree vertices (ids >=
    // hub_base) as a dense bitmap, row v - hub_base, bit c - hub_base set iff
    // v -> c.  Membership tests between two hubs are one load instead of a
    // binary search in a long list.
    int32_t hub_base;       // n when there is no hub matrix
    int32_t hub_wpr;        // 32-bit words per row
    const uint32_t *hub;
    // Summary of the hub matrix, one bit per 32-bit word of it (set iff the word is not zero), hub_cwpr
    // words per row: 1/32 of the matrix, small enough to stay in the caches.  Nine words in ten of the
    // matrix are zero, so a kernel that tests scattered pairs asks the summary first and fetches a line of
    // the matrix itself only for the words that hold something.  nullptr: no summary.
    // Columns from hub_dense on (the highest-degree vertices, where most words hold a bit) are read from the
    // matrix directly; 0 without a summary.
    int32_t hub_cwpr;
    int32_t hub_dense;
    const uint32_t *hubc;
    // labcnt[v * nlab + l] = neighbours of v that carry label l (symmetric
    // labelled graphs only, else nullptr): lets a leaf level be counted
    // without walking the list.
    int32_t nlab;
    const uint32_t *labcnt;
    // Label-grouped copy of the lists (symmetric labelled graphs only, else
    // nullptr): v's neighbours ordered by (label, id); labseg[v * (nlab + 1) + l]
    // is where label l starts inside v's list.  A level that asks for a label
    // takes its candidates from that segment alone instead of filtering the
    // whole list; membership tests keep using the id-ordered lists.
    const int32_t *adjL;
    const uint32_t *labseg;
    // The arrays of one direction, chosen at run time.  A select between the two kernel arguments: indexing
    // the pair with a run-time value makes the compiler keep a copy of it in private memory, and every
    // membership test would start with a load from there.
    __device__ __forceinline__ const int64_t *offs(int dir) const { return dir ? off[1] : off[0]; }
    __device__ __forceinline__ const int32_t *adjs(int dir) const { return dir ? adj[1] : adj[0]; }
    __device__ __forceinline__ const int64_t *mids(int dir) const { return dir ? mid[1] : mid[0]; }
};

__device__ __forceinline__ bool hub_arc(const GraphView &g, int32_t from, int32_t to) {
    uint32_t c = (uint32_t)(to - g.hub_base);
    return (g.hub[(size_t)(from - g.hub_base) * (size_t)g.hub_wpr + (c >> 5)] >> (c & 31)) & 1u;
}

enum { PART_FULL = 0, PART_ABOVE = 1, PART_BELOW = 2 };

// One level of the matching plan, flattened for the device.
struct LevelPlan {
    int8_t src_level, src_dir, src_part;
    int8_t n_cons, n_lower, n_upper;
    int8_t cons_level[GPM_MAXC], cons_dir[GPM_MAXC], cons_part[GPM_MAXC];
    int8_t lower_level[GPM_MAXC], upper_level[GPM_MAXC];
    int8_t lower_is_src;  // the only lower bound is the source owner (range = PART_ABOVE, no search)
    uint16_t src_adj;     // levels whose pattern vertex is adjacent to this level's source vertex: their data
                          // vertices are known neighbours of the source (symmetric graphs), no search needed
    int32_t label, min_out, min_in;
};

struct PlanView {
    int32_t P;
    int8_t simple;           // data graph without duplicate arcs: any operand list may be the one walked
    int8_t order[GPM_MAXP];  // pattern vertex bound at each level
    LevelPlan lv[GPM_MAXP];
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (GPM_WAVE - 1); }

// First position in [b, e) of a sorted array whose value is >= key.
__device__ __forceinline__ int64_t lower_bound_g(const int32_t *a, int64_t b, int64_t e, int32_t key) {
    while (b < e) {
        int64_t m = b + ((e - b) >> 1);
        if (a[m] < key) b = m + 1;
        else e = m;
    }
    return b;
}

__device__ __forceinline__ bool contains_g(const int32_t *a, int64_t b, int64_t e, int32_t key) {
    int64_t p = lower_bound_g(a, b, e, key);
    return p < e && a[p] == key;
}

// Same on an LDS-resident sorted array with 32-bit positions.
__device__ __forceinline__ int lower_bound_s(const int32_t *a, int b, int e, int32_t key) {
    while (b < e) {
        int m = (b + e) >> 1;
        if (a[m] < key) b = m + 1;
        else e = m;
    }
    return b;
}

// Is c in N_dir(v)?  (dir 0: arc v -> c, dir 1: arc c -> v.)  `part` only
// narrows the search; the caller guarantees c lies in that part.
__device__ __forceinline__ bool in_list(const GraphView &g, int dir, int part, int32_t v, int32_t c);

__device__ __forceinline__ void list_range(const GraphView &g, int dir, int part, int32_t v, int64_t &b, int64_t &e) {
    // written with selects: two assignments under conditions get merged into one store at a run-time
    // offset, which puts b and e into private memory
    const int64_t lo = g.offs(dir)[v], hi = g.offs(dir)[v + 1];
    const int64_t md = part == PART_FULL ? lo : g.mids(dir)[v];
    b = part == PART_ABOVE ? md : lo;
    e = part == PART_BELOW ? md : hi;
}

__device__ __forceinline__ bool in_list(const GraphView &g, int dir, int part, int32_t v, int32_t c) {
    if (v >= g.hub_base && c >= g.hub_base) return dir == 0 ? hub_arc(g, v, c) : hub_arc(g, c, v);
    int64_t b, e;
    list_range(g, dir, part, v, b, e);
    return contains_g(g.adjs(dir), b, e, c);
}

// Does this level draw its candidates from a label segment?
__device__ __forceinline__ bool from_segment(const GraphView &g, const LevelPlan &lv) {
    return lv.label >= 0 && g.adjL != nullptr;
}

// Candidate range of one level for the partial match m: positions [b, e) of
// `arr`, the source list (or its segment of the level's label) cut down by the
// ordering bounds (c > every lower bound, c < every upper bound).
// Bounds that refer to level `skip_level` are left out (that vertex is not bound yet).
__device__ __forceinline__ void source_range(const GraphView &g, const LevelPlan &lv, const int32_t *m, int64_t &b,
                                             int64_t &e, const int32_t *&arr, int skip_level = -1) {
    if (from_segment(g, lv)) {
        const int32_t v = m[lv.src_level];
        arr = g.adjL;
        b = e = 0;
        if (lv.label >= g.nlab) return;  // nobody carries this label
        const uint32_t *seg = g.labseg + (size_t)v * (size_t)(g.nlab + 1) + (size_t)lv.label;
        b = g.off[0][v] + seg[0];
        e = g.off[0][v] + seg[1];
        int32_t lo = lv.src_part == PART_ABOVE ? v : -1, hi = lv.src_part == PART_BELOW ? v : INT32_MAX;
        for (int j = 0; j < lv.n_lower; ++j)
            if (lv.lower_level[j] != skip_level) lo = max(lo, m[lv.lower_level[j]]);
        for (int j = 0; j < lv.n_upper; ++j)
            if (lv.upper_level[j] != skip_level) hi = min(hi, m[lv.upper_level[j]]);
        if (lo >= 0) b = lower_bound_g(arr, b, e, lo + 1);
        if (hi != INT32_MAX) e = lower_bound_g(arr, b, e, hi);
        return;
    }
    arr = g.adjs(lv.src_dir);
    list_range(g, lv.src_dir, lv.src_part, m[lv.src_level], b, e);
    if (lv.n_lower && !lv.lower_is_src) {
        int32_t lo = -1;
        for (int j = 0; j < lv.n_lower; ++j)
            if (lv.lower_level[j] != skip_level) lo = max(lo, m[lv.lower_level[j]]);
        if (lo >= 0) b = lower_bound_g(g.adjs(lv.src_dir), b, e, lo + 1);
    }
    if (lv.n_upper) {
        int32_t hi = INT32_MAX;
        for (int j = 0; j < lv.n_upper; ++j)
            if (lv.upper_level[j] != skip_level) hi = min(hi, m[lv.upper_level[j]]);
        if (hi != INT32_MAX) e = lower_bound_g(g.adjs(lv.src_dir), b, e, hi);
    }
}

// The list a level walks for the partial match m: the source list, or on a
// simple graph the shortest of the source and constraint lists (the result set
// is the same), cut down exactly by the ordering bounds.  walk = -1 for the
// source list, else the index of the constraint whose list is walked.
__device__ __forceinline__ void walk_range(const GraphView &g, const PlanView &pl, const LevelPlan &lv,
                                           const int32_t *m, int64_t &b, int64_t &e, const int32_t *&arr, int &walk) {
    source_range(g, lv, m, b, e, arr);
    walk = -1;
    if (!pl.simple || lv.n_cons == 0) return;
    int64_t best = e - b;
    for (int j = 0; j < lv.n_cons; ++j) {
        int64_t cb, ce;
        list_range(g, lv.cons_dir[j], lv.cons_part[j], m[lv.cons_level[j]], cb, ce);
        if (ce - cb < best) {
            best = ce - cb;
            b = cb;
            e = ce;
            walk = j;
        }
    }
    if (walk < 0) return;
    arr = g.adjs(lv.cons_dir[walk]);
    int32_t lo = -1, hi = INT32_MAX;
    for (int j = 0; j < lv.n_lower; ++j) lo = max(lo, m[lv.lower_level[j]]);
    for (int j = 0; j < lv.n_upper; ++j) hi = min(hi, m[lv.upper_level[j]]);
    if (lo >= 0) b = lower_bound_g(arr, b, e, lo + 1);
    if (hi != INT32_MAX) e = lower_bound_g(arr, b, e, hi);
}

// Per-vertex filters of a level: label and minimum degrees.
__device__ __forceinline__ bool vertex_ok(const GraphView &g, const LevelPlan &lv, int32_t c) {
    if (lv.label >= 0 && g.label[c] != (uint8_t)lv.label) return false;
    if (lv.min_out > 0 && g.off[0][c + 1] - g.off[0][c] < lv.min_out) return false;
    if (lv.min_in > 0 && g.off[1][c + 1] - g.off[1][c] < lv.min_in) return false;
    return true;
}

// Full acceptance test of candidate c at a level whose partial match is
// m[0..k): not already used, vertex filters, and membership in the list of
// every other bound pattern neighbour.  skip_cons >= 0 leaves that
// constraint out and tests the source list instead (used when the caller
// walks that constraint's list rather than the source list).
__device__ __forceinline__ bool accept(const GraphView &g, const LevelPlan &lv, const int32_t *m, int k, int32_t c,
                                       int skip_cons = -1) {
    for (int t = 0; t < k; ++t)
        if (m[t] == c) return false;
    if (lv.label >= 0 && g.label[c] != (uint8_t)lv.label) return false;  // one byte, very selective: first
    for (int j = 0; j < lv.n_cons; ++j) {
        bool in = j == skip_cons ? in_list(g, lv.src_dir, lv.src_part, m[lv.src_level], c)
                                 : in_list(g, lv.cons_dir[j], lv.cons_part[j], m[lv.cons_level[j]], c);
        if (!in) return false;
    }
    // the degree filter rarely rejects a vertex that passed the membership tests: last
    if (lv.min_out > 0 && g.off[0][c + 1] - g.off[0][c] < lv.min_out) return false;
    if (lv.min_in > 0 && g.off[1][c + 1] - g.off[1][c] < lv.min_in) return false;
    return true;
}

// Merge-model bytes of extending one partial match of length k at this
// level: the match itself plus one pass over every operand list, each list
// counted at its ordered length.
// sub_level >= 0: the vertex of that level is sub_v instead of m[sub_level].
__device__ __forceinline__ uint64_t level_bytes(const GraphView &g, const LevelPlan &lv, const int32_t *m, int k,
                                                int sub_level = -1, int32_t sub_v = 0) {
    int64_t b, e;
    list_range(g, lv.src_dir, lv.src_part, lv.src_level == sub_level ? sub_v : m[lv.src_level], b, e);
    if (from_segment(g, lv)) {  // only the label's segment of the source list is streamed
        b = e = 0;
        if (lv.label < g.nlab) {
            const int32_t sv = lv.src_level == sub_level ? sub_v : m[lv.src_level];
            const uint32_t *seg = g.labseg + (size_t)sv * (size_t)(g.nlab + 1) + (size_t)lv.label;
            e = (int64_t)seg[1] - (int64_t)seg[0];
        }
    }
    uint64_t bytes = 4ull * (uint64_t)k + 4ull * (uint64_t)(e - b);
    for (int j = 0; j < lv.n_cons; ++j) {
        list_range(g, lv.cons_dir[j], lv.cons_part[j], lv.cons_level[j] == sub_level ? sub_v : m[lv.cons_level[j]], b, e);
        bytes += 4ull * (uint64_t)(e - b);
    }
    return bytes;
}

// Number of candidates of a leaf level -- one whose only requirement besides
// the label is adjacency to its source vertex s -- for a partial match m[0..k)
// on a simple symmetric graph: the (label-matching) neighbours of s that are
// not already in the match.  No list is walked.
__device__ __forceinline__ int64_t leaf_count(const GraphView &g, const LevelPlan &lv, const int32_t *m, int k,
                                              int32_t s) {
    int64_t c = lv.label >= 0 ? (int64_t)g.labcnt[(size_t)s * (size_t)g.nlab + (size_t)lv.label]
                              : g.off[0][s + 1] - g.off[0][s];
    for (int t = 0; t < k; ++t) {
        const int32_t x = m[t];
        if (x != s && (lv.label < 0 || g.label[x] == (uint8_t)lv.label) &&
            (((lv.src_adj >> t) & 1u) || in_list(g, 0, PART_FULL, s, x)))
            c -= 1;
    }
    return c;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, GPM_WAVE);
    return v;  // valid in lane 0
}

// Orders a wave's own LDS stores before its later LDS loads.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
