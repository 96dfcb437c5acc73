// Hand-written gfx950 kernels of the match search.
//
// Arithmetic is 32-bit id compares and 64-bit counts; the work is irregular
// integer set intersection, bounded by memory traffic, so there is no MFMA
// anywhere.  Wave width is 64 throughout.
#pragma once
#include "gpm_device.hpp"

// ---------------------------------------------------------------------------
// Graph preparation (runs once per upload)
// ---------------------------------------------------------------------------

// Total degree of every vertex (sort key of the renumbering).
__global__ void k_total_degree(int32_t n, const int64_t *out_off, const int64_t *in_off, uint32_t *deg, int32_t *ids) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    int64_t d = out_off[v + 1] - out_off[v];
    if (in_off) d += in_off[v + 1] - in_off[v];
    else d *= 2;
    deg[v] = (uint32_t)min(d, (int64_t)0xffffffffll);
    ids[v] = v;
}

// rank[old] = new from perm[new] = old, and the new degree array.
__global__ void k_scatter_rank(int32_t n, const int32_t *perm, int32_t *rank, const int64_t *old_off, int64_t *new_deg) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    int32_t o = perm[v];
    rank[o] = v;
    new_deg[v] = old_off[o + 1] - old_off[o];
}

// One 64-bit sort key per arc: (new source id << 32) | new target id.
__global__ void k_arc_keys(int32_t n, int64_t m, const int64_t *off, const int32_t *adj, const int32_t *rank,
                           uint64_t *keys) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    int32_t lo = 0, hi = n;  // last v with off[v] <= e
    while (hi - lo > 1) {
        int32_t mid = lo + ((hi - lo) >> 1);
        if (off[mid] <= e) lo = mid;
        else hi = mid;
    }
    keys[e] = ((uint64_t)(uint32_t)rank[lo] << 32) | (uint32_t)rank[adj[e]];
}

// Unpack the sorted keys into the adjacency array; one thread per arc also raises flag bit 0 for a self loop
// (source == target) and for a duplicate arc (the key equals the one before it).
__global__ void k_keys_to_adj(int64_t m, const uint64_t *keys, int32_t *adj, uint32_t *flags) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m) return;
    const uint64_t key = keys[e];
    adj[e] = (int32_t)(uint32_t)key;
    if ((uint32_t)(key >> 32) == (uint32_t)key || (e > 0 && keys[e - 1] == key)) atomicOr(flags, 1u);
}

// mid[v] = first position of v's list holding an id > v.
__global__ void k_mid(int32_t n, const int64_t *off, const int32_t *adj, int64_t *mid) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    mid[v] = lower_bound_g(adj, off[v], off[v + 1], v + 1);
}

// Validation of caller-supplied CSR before anything dereferences it.
__global__ void k_validate_adj(int32_t n, int64_t m, const int32_t *adj, uint32_t *flags) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < m && (adj[e] < 0 || adj[e] >= n)) atomicOr(flags, 2u);
}

__global__ void k_validate_off(int32_t n, int64_t m, const int64_t *off, uint32_t *flags) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > n) return;
    bool bad = off[v] < 0 || off[v] > m || (v < n && off[v] > off[v + 1]) || (v == 0 && off[0] != 0) ||
               (v == n && off[n] != m);
    if (bad) atomicOr(flags, 2u);
}

__global__ void k_permute_labels(int32_t n, const int32_t *perm, const uint8_t *in, uint8_t *out) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) out[v] = in[perm[v]];
}

// ---------------------------------------------------------------------------
// Level 0: label / degree candidate filter over a chunk of roots
// ---------------------------------------------------------------------------

// Roots are handed out heaviest first: root index i is vertex n-1-i.  Each
// wave compacts its survivors with one ballot and one atomic.
__global__ void k_filter_roots(GraphView g, LevelPlan lv0, int64_t lo, int64_t hi, int32_t *rows,
                               unsigned long long *n_rows) {
    int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    int32_t v = 0;
    if (i < hi) {
        v = (int32_t)(g.n - 1 - i);
        ok = vertex_ok(g, lv0, v);
    }
    unsigned long long mask = __ballot(ok);
    if (!mask) return;
    int lane = lane_id();
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(n_rows, (unsigned long long)__popcll(mask));
    base = __shfl(base, 0, GPM_WAVE);
    if (ok) rows[base + __popcll(mask & ((1ull << lane) - 1))] = v;
}

// ---------------------------------------------------------------------------
// Level k: candidate ranges, then load-balanced extension
// ---------------------------------------------------------------------------

// Per partial match: where its candidates start in the adjacency array and
// how many there are (after the ordering bounds).  Also sums the merge-model
// bytes of this level.
__global__ void k_ranges(GraphView g, PlanView pl, int k, const int32_t *rows, int64_t nrows, int64_t *beg,
                         int64_t *cnt, int8_t *wsel, unsigned long long *alg_bytes) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t bytes = 0;
    if (r < nrows) {
        const int32_t *m = rows + r * k;
        const LevelPlan &lv = pl.lv[k];
        int64_t b, e;
        const int32_t *arr;
        int walk;
        walk_range(g, pl, lv, m, b, e, arr, walk);
        beg[r] = b;
        cnt[r] = e - b;
        wsel[r] = (int8_t)walk;
        bytes = level_bytes(g, lv, m, k);
    }
    bytes = wave_sum_u64(bytes);
    if (lane_id() == 0 && bytes) atomicAdd(alg_bytes, (unsigned long long)bytes);
}

// Largest b1 in (b0, nrows] such that the candidates of rows [b0, b1) fit in
// `cap`; at least one row is always taken.
__global__ void k_batch_end(const int64_t *pref, int64_t nrows, int64_t b0, int64_t cap, int64_t *out) {
    int64_t base = pref[b0];
    int64_t lo = b0 + 1, hi = nrows;  // answer in [lo, hi]
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo + 1) >> 1);
        if (pref[mid] - base <= cap) lo = mid;
        else hi = mid - 1;
    }
    out[0] = lo;
    out[1] = pref[lo] - base;
}

enum { MODE_EXTEND = 0, MODE_COUNT = 1, MODE_EMIT = 2 };

// One thread per candidate of the concatenated candidate space [g0, g1) of
// rows [r0, r1): perfectly balanced whatever the degree skew.  Survivors are
// compacted per wave by ballot + prefix popcount and one atomic.
//   MODE_EXTEND: survivors become rows of k+1 ids in `out`
//   MODE_COUNT : survivors are only counted (last level, count-only)
//   MODE_EMIT  : survivors complete a match, written in pattern-vertex order
//                with the caller's 1-based ids
template <int MODE>
__global__ void k_extend(GraphView g, PlanView pl, int k, const int32_t *rows, const int64_t *pref, const int64_t *beg,
                         const int8_t *wsel, int64_t r0, int64_t r1, int64_t g0, int64_t g1, int32_t *out,
                         unsigned long long *n_out, const int32_t *perm) {
    int64_t gi = g0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    int32_t c = 0;
    const int32_t *m = rows;
    if (gi < g1) {
        int64_t lo = r0, hi = r1;  // last row with pref[row] <= gi
        while (hi - lo > 1) {
            int64_t mid = lo + ((hi - lo) >> 1);
            if (pref[mid] <= gi) lo = mid;
            else hi = mid;
        }
        const LevelPlan &lv = pl.lv[k];
        m = rows + lo * k;
        const int walk = wsel[lo];  // which operand list this row walks (see walk_range)
        const int32_t *arr = walk >= 0 ? g.adjs(lv.cons_dir[walk]) : from_segment(g, lv) ? g.adjL : g.adjs(lv.src_dir);
        c = arr[beg[lo] + (gi - pref[lo])];
        ok = accept(g, lv, m, k, c, walk);
    }
    unsigned long long mask = __ballot(ok);
    if (!mask) return;
    int lane = lane_id();
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(n_out, (unsigned long long)__popcll(mask));
    if (MODE == MODE_COUNT) return;
    base = __shfl(base, 0, GPM_WAVE);
    if (!ok) return;
    unsigned long long row = base + __popcll(mask & ((1ull << lane) - 1));
    if (MODE == MODE_EXTEND) {
        int32_t *o = out + row * (k + 1);
        for (int t = 0; t < k; ++t) o[t] = m[t];
        o[k] = c;
    } else {
        int32_t *o = out + row * pl.P;
        for (int t = 0; t < k; ++t) o[pl.order[t]] = perm[m[t]] + 1;
        o[pl.order[k]] = perm[c] + 1;
    }
}

// Single-vertex pattern in output mode: every surviving root is a match.
__global__ void k_emit_roots(int64_t nrows, const int32_t *rows, const int32_t *perm, int32_t *out) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrows) out[r] = perm[rows[r]] + 1;
}

// ---------------------------------------------------------------------------
// Fused last two levels: neighbour-list intersection staged through LDS
// ---------------------------------------------------------------------------

#define TAIL_WAVES 4
#define TAIL_CAP 1024  // ids of one wave's staged candidate set (4 KiB)

// How the second-to-last level (A) and the last level (B) relate.  When
// `reuse` is set, B's operand lists are A's lists plus the lists owned by the
// A vertex itself, so B's candidates are (A's survivor set) intersected with
// those extra lists: the staged set is probed in LDS instead of re-reading
// A's lists from memory for every A vertex.
struct TailInfo {
    int8_t reuse;
    int8_t n_extra;         // lists of level B owned by the A vertex (1 or 2)
    int8_t extra_dir[2], extra_part[2];
    int8_t b_lower_is_a;    // B requires c > a (a is B's largest lower bound)
    int8_t b_upper_is_a;    // B requires c < a
    int8_t any_shortest;    // simple graph: level A may walk its shortest list
    int8_t leaf_b;          // level B is a leaf level: counted in closed form per A survivor
    int8_t leaf_pair;       // levels A and B are both leaves off earlier vertices: closed form per row
    int8_t pair_same_set;   // ... and draw from the same candidate set (same source, same label)
    int8_t indep;           // level B has no list of the A vertex: its candidate set does not depend on a, so
                            // it is walked once per staged set and paired with it (|A||B| minus the clashes)
    int8_t twin;            // level B has exactly level A's lists, bounds and filters and no list of the A
                            // vertex: the two draw from one survivor set S, counted as pairs of S
};

// Count |{c in staged[sb, se) : c in every extra list of a, c != a}|, where
// the wave either walks the staged set and probes a's list in memory, or
// streams a's list with coalesced reads and probes the staged set in LDS,
// whichever side is shorter.
__device__ __forceinline__ uint32_t tail_reuse_count(const GraphView &g, const TailInfo &ti, const int32_t *staged,
                                                     int sb, int se, int32_t a, int lane) {
    uint32_t cnt = 0;
    int64_t b0, e0;
    list_range(g, ti.extra_dir[0], ti.extra_part[0], a, b0, e0);
    const int32_t *adj0 = g.adjs(ti.extra_dir[0]);
    int64_t b1 = 0, e1 = 0;
    const int32_t *adj1 = nullptr;
    if (ti.n_extra > 1) {
        list_range(g, ti.extra_dir[1], ti.extra_part[1], a, b1, e1);
        adj1 = g.adjs(ti.extra_dir[1]);
    }
    if ((int64_t)(se - sb) <= e0 - b0) {
        for (int i = sb + lane; i < se; i += GPM_WAVE) {
            int32_t c = staged[i];
            bool ok = c != a && in_list(g, ti.extra_dir[0], ti.extra_part[0], a, c);
            if (ok && adj1) ok = in_list(g, ti.extra_dir[1], ti.extra_part[1], a, c);
            cnt += ok;
        }
    } else {
        // narrow a's list to the id window of the staged range first
        int32_t first = staged[sb], last = staged[se - 1];
        b0 = lower_bound_g(adj0, b0, e0, first);
        e0 = lower_bound_g(adj0, b0, e0, last + 1);
        for (int64_t i = b0 + lane; i < e0; i += GPM_WAVE) {
            int32_t c = adj0[i];
            int p = lower_bound_s(staged, sb, se, c);
            bool ok = p < se && staged[p] == c && c != a;
            if (ok && adj1) ok = contains_g(adj1, b1, e1, c);
            cnt += ok;
        }
    }
    return cnt;
}

// One wave per partial match of length L = P-2.  Level A survivors are
// compacted into the wave's LDS slice (ballot + prefix popcount); level B is
// evaluated per survivor and only counted.
__global__ __launch_bounds__(TAIL_WAVES *GPM_WAVE) void k_tail(GraphView g, PlanView pl, TailInfo ti,
                                                               const int32_t *rows, int64_t nrows,
                                                               unsigned long long *n_matches,
                                                               unsigned long long *alg_bytes) {
    __shared__ int32_t s_set[TAIL_WAVES][TAIL_CAP];
    __shared__ int32_t s_m[TAIL_WAVES][GPM_MAXP + 2];
    const int wave = threadIdx.x / GPM_WAVE, lane = lane_id();
    const int L = pl.P - 2;
    const LevelPlan &la = pl.lv[L];
    const LevelPlan &lb = pl.lv[L + 1];
    int32_t *staged = s_set[wave];
    int32_t *m = s_m[wave];
    uint64_t cnt = 0, bytes = 0;

    for (int64_t r = (int64_t)blockIdx.x * TAIL_WAVES + wave; r < nrows; r += (int64_t)gridDim.x * TAIL_WAVES) {
        wave_lds_sync();
        if (lane < L) m[lane] = rows[r * L + lane];
        wave_lds_sync();
        if (lane == 0) bytes += level_bytes(g, la, m, L);

        // Level A: walk the source list, or on a simple graph whichever
        // operand list is shortest (the result set is the same).  The
        // ordering bounds cut the source range exactly; on another list they
        // are tested per element.
        int walk = -1;  // -1 = source list, j = constraint j
        int64_t b, e;
        const int32_t *sarr;
        source_range(g, la, m, b, e, sarr);
        int32_t lo_id = -1, hi_id = INT32_MAX;
        if (ti.any_shortest) {
            for (int j = 0; j < la.n_cons; ++j) {
                int64_t cb, ce;
                list_range(g, la.cons_dir[j], la.cons_part[j], m[la.cons_level[j]], cb, ce);
                if (ce - cb < e - b) {
                    b = cb;
                    e = ce;
                    walk = j;
                }
            }
            if (walk >= 0) {
                for (int j = 0; j < la.n_lower; ++j) lo_id = max(lo_id, m[la.lower_level[j]]);
                for (int j = 0; j < la.n_upper; ++j) hi_id = min(hi_id, m[la.upper_level[j]]);
            }
        }
        const int32_t *wadj = walk < 0 ? sarr : g.adjs(la.cons_dir[walk]);
        if (walk >= 0 && e - b > GPM_WAVE) {  // a long list is entered and left at the bounds
            if (lo_id >= 0) b = lower_bound_g(wadj, b, e, lo_id + 1);
            if (hi_id != INT32_MAX) e = lower_bound_g(wadj, b, e, hi_id);
        }
        if (ti.twin) {
            // levels A and B are twins: |S| survivors give |S|(|S|-1) ordered pairs, or
            // half of that when an ordering bound ties B to A; nothing is staged
            uint32_t s_cnt = 0;
            for (int64_t i = b + lane; i < e; i += GPM_WAVE) {
                const int32_t c = wadj[i];
                s_cnt += c > lo_id && c < hi_id && accept(g, la, m, L, c, walk);
            }
            const uint64_t s_all = wave_sum_u64(s_cnt);
            if (lane == 0) {
                const uint64_t pairs = s_all * (s_all - (s_all ? 1 : 0));
                cnt += (ti.b_lower_is_a || ti.b_upper_is_a) ? pairs / 2 : pairs;
                bytes += s_all * (4ull + level_bytes(g, la, m, L));  // level B per survivor: one more id, same lists
            }
            continue;
        }
        const bool reuse_row = ti.reuse && (e - b) <= TAIL_CAP;

        int n_staged = 0;
        for (int64_t i0 = b; i0 < e; i0 += GPM_WAVE) {
            int64_t i = i0 + lane;
            bool ok = false;
            int32_t c = 0;
            if (i < e) {
                c = wadj[i];
                ok = c > lo_id && c < hi_id && accept(g, la, m, L, c, walk);
            }
            unsigned long long mask = __ballot(ok);
            if (ok) staged[n_staged + __popcll(mask & ((1ull << lane) - 1))] = c;
            n_staged += __popcll(mask);
            bool last = i0 + GPM_WAVE >= e;
            if (!last && (reuse_row || n_staged + GPM_WAVE <= TAIL_CAP)) continue;
            wave_lds_sync();
            if (ti.leaf_b && !reuse_row) {
                // level B is a leaf: one lane per survivor a, nothing is walked
                for (int ia = lane; ia < n_staged; ia += GPM_WAVE) {
                    const int32_t a = staged[ia];
                    const int32_t s = lb.src_level == L ? a : m[lb.src_level];
                    int64_t c = leaf_count(g, lb, m, L, s);
                    if (a != s && (lb.label < 0 || g.label[a] == (uint8_t)lb.label) &&
                        (((lb.src_adj >> L) & 1u) || in_list(g, 0, PART_FULL, s, a)))
                        c -= 1;
                    if (c > 0) cnt += (uint64_t)c;
                    bytes += 4ull * (uint64_t)(L + 1) + 4ull * (uint64_t)(g.off[0][s + 1] - g.off[0][s]);
                }
                n_staged = 0;
                wave_lds_sync();
                continue;
            }
            if (ti.indep && n_staged == 0) continue;
            if (ti.indep) {
                // level B does not depend on a: walk its candidates once and pair each with the staged
                // (ascending) survivors of level A that it may go with
                int wb = -1;
                int64_t b2, e2;
                const int32_t *barr;
                source_range(g, lb, m, b2, e2, barr, L);
                for (int j = 0; j < lb.n_cons; ++j) {
                    int64_t cb, ce;
                    list_range(g, lb.cons_dir[j], lb.cons_part[j], m[lb.cons_level[j]], cb, ce);
                    if (ce - cb < e2 - b2) {
                        b2 = cb;
                        e2 = ce;
                        wb = j;
                    }
                }
                int32_t lo2 = -1, hi2 = INT32_MAX;
                if (wb >= 0) {
                    barr = g.adjs(lb.cons_dir[wb]);
                    for (int j = 0; j < lb.n_lower; ++j)
                        if (lb.lower_level[j] != L) lo2 = max(lo2, m[lb.lower_level[j]]);
                    for (int j = 0; j < lb.n_upper; ++j)
                        if (lb.upper_level[j] != L) hi2 = min(hi2, m[lb.upper_level[j]]);
                }
                for (int64_t i2 = b2 + lane; i2 < e2; i2 += GPM_WAVE) {
                    const int32_t c2 = barr[i2];
                    if (!(c2 > lo2 && c2 < hi2 && accept(g, lb, m, L, c2, wb))) continue;
                    const int p = lower_bound_s(staged, 0, n_staged, c2);  // survivors below c2
                    const int same = p < n_staged && staged[p] == c2;
                    cnt += ti.b_lower_is_a ? (uint32_t)p : ti.b_upper_is_a ? (uint32_t)(n_staged - p - same)
                                                                          : (uint32_t)(n_staged - same);
                }
                if (lane == 0) {  // level B per survivor: its ids and one pass over B's lists (none owned by a)
                    m[L] = staged[0];
                    bytes += (uint64_t)n_staged * level_bytes(g, lb, m, L + 1);
                }
                n_staged = 0;
                wave_lds_sync();
                continue;
            }
            // level B for every staged survivor a
            for (int ia = 0; ia < n_staged; ++ia) {
                int32_t a = staged[ia];
                if (reuse_row) {
                    int sb = ti.b_lower_is_a ? ia + 1 : 0;
                    int se = ti.b_upper_is_a ? ia : n_staged;
                    m[L] = a;  // same value from every lane
                    wave_lds_sync();
                    if (lane == 0) bytes += level_bytes(g, lb, m, L + 1);
                    if (se > sb) cnt += tail_reuse_count(g, ti, staged, sb, se, a, lane);
                } else {
                    m[L] = a;
                    wave_lds_sync();
                    if (lane == 0) bytes += level_bytes(g, lb, m, L + 1);
                    int64_t b2, e2;
                    const int32_t *badj;
                    source_range(g, lb, m, b2, e2, badj);
                    for (int64_t i2 = b2 + lane; i2 < e2; i2 += GPM_WAVE) cnt += accept(g, lb, m, L + 1, badj[i2]);
                }
            }
            n_staged = 0;
            wave_lds_sync();
        }
    }
    cnt = wave_sum_u64(cnt);
    bytes = wave_sum_u64(bytes);
    if (lane == 0) {
        if (cnt) atomicAdd(n_matches, (unsigned long long)cnt);
        if (bytes) atomicAdd(alg_bytes, (unsigned long long)bytes);
    }
}

// Twin last levels (TailInfo::twin) with G lanes per partial match instead of a
// whole wave: the lists walked here are mostly far shorter than 64 entries, so
// 64/G partial matches share a wave.  Nothing is staged: |S| survivors of level
// A give |S|(|S|-1) ordered pairs, or half of that under an ordering bound
// between the twins.
template <int G>
__global__ __launch_bounds__(256) void k_tail_twin(GraphView g, PlanView pl, TailInfo ti, const int32_t *rows,
                                                   int64_t nrows, unsigned long long *n_matches,
                                                   unsigned long long *alg_bytes) {
    const int lane = lane_id(), sub = lane & (G - 1);
    const int L = pl.P - 2;
    const LevelPlan &la = pl.lv[L];
    const int64_t groups = (int64_t)gridDim.x * (blockDim.x / G);
    uint64_t cnt = 0, bytes = 0;
    for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G; r < nrows; r += groups) {
        const int32_t *m = rows + r * L;
        int walk = -1;
        int64_t b, e;
        const int32_t *sarr;
        source_range(g, la, m, b, e, sarr);
        int32_t lo_id = -1, hi_id = INT32_MAX;
        for (int j = 0; j < la.n_cons; ++j) {  // twins only exist on simple graphs: any list may be walked
            int64_t cb, ce;
            list_range(g, la.cons_dir[j], la.cons_part[j], m[la.cons_level[j]], cb, ce);
            if (ce - cb < e - b) {
                b = cb;
                e = ce;
                walk = j;
            }
        }
        if (walk >= 0) {
            for (int j = 0; j < la.n_lower; ++j) lo_id = max(lo_id, m[la.lower_level[j]]);
            for (int j = 0; j < la.n_upper; ++j) hi_id = min(hi_id, m[la.upper_level[j]]);
        }
        const int32_t *wadj = walk < 0 ? sarr : g.adjs(la.cons_dir[walk]);
        if (walk >= 0 && e - b > 4 * G) {  // a long list is entered and left at the bounds
            if (lo_id >= 0) b = lower_bound_g(wadj, b, e, lo_id + 1);
            if (hi_id != INT32_MAX) e = lower_bound_g(wadj, b, e, hi_id);
        }
        uint32_t s_cnt = 0;
        for (int64_t i = b + sub; i < e; i += G) {
            const int32_t c = wadj[i];
            s_cnt += c > lo_id && c < hi_id && accept(g, la, m, L, c, walk);
        }
        for (int o = G / 2; o > 0; o >>= 1) s_cnt += __shfl_xor(s_cnt, o, GPM_WAVE);
        if (sub == 0) {
            const uint64_t s_all = s_cnt;
            const uint64_t pairs = s_all * (s_all - (s_all ? 1 : 0));
            cnt += (ti.b_lower_is_a || ti.b_upper_is_a) ? pairs / 2 : pairs;
            bytes += (s_all + 1) * level_bytes(g, la, m, L) + 4ull * s_all;
        }
    }
    cnt = wave_sum_u64(cnt);
    bytes = wave_sum_u64(bytes);
    if (lane == 0) {
        if (cnt) atomicAdd(n_matches, (unsigned long long)cnt);
        if (bytes) atomicAdd(alg_bytes, (unsigned long long)bytes);
    }
}

// Three fused levels: level Z = P-3 followed by twin last levels whose lists are
// Z's lists, the list of the Z vertex itself and possibly lists of further
// earlier vertices (K6 minus a perfect matching: z, then two twins adjacent to
// z and to everything z is adjacent to).  For one partial match of length P-3
// the common part T of Z's lists is walked once; its members that pass Z's
// bounds and filters are the candidates z, its members that pass the twins'
// bounds, filters and further lists are the set B the twins draw from whatever
// z is, and every z then only needs |B ^ N(z)|.  This replaces one extension
// level and its frontier, and the recomputation of T for every z.  G lanes per
// partial match and sets of at most CAP vertices; partial matches with larger
// sets are handed back (`over`): first to a launch with whole waves and larger
// sets, then to the level-by-level path.  `sel` (optional) lists the partial
// matches to take.
struct Tail3Info {
    int8_t enabled;
    int8_t n_zl, zl[GPM_MAXC + 1];  // levels whose lists level Z intersects
    int8_t n_xl, xl[GPM_MAXC + 1];  // further levels whose lists the twins need
};

__device__ __forceinline__ bool t3_level_ok(const GraphView &g, const LevelPlan &lv, const int32_t *m, int32_t c) {
    for (int j = 0; j < lv.n_lower; ++j)
        if (c <= m[lv.lower_level[j]]) return false;
    for (int j = 0; j < lv.n_upper; ++j)
        if (c >= m[lv.upper_level[j]]) return false;
    return vertex_ok(g, lv, c);
}

// `nrows_dev` (optional): the number of entries of `sel` sits in device memory (the hand-back count of the
// launch before), so that a tier can be enqueued without the host reading that count first.
template <int G, int CAP>
__global__ __launch_bounds__(256) void k_tail3(GraphView g, PlanView pl, TailInfo ti, Tail3Info t3, const int32_t *rows,
                                               const int64_t *sel, int64_t nrows, const unsigned long long *nrows_dev,
                                               int64_t *over, unsigned long long *n_over, unsigned long long *n_matches,
                                               unsigned long long *alg_bytes) {
    if (nrows_dev) nrows = (int64_t)*nrows_dev;
    __shared__ int32_t s_z[256 / G][CAP], s_b[256 / G][CAP];
    const int lane = lane_id(), sub = lane & (G - 1), shift = lane & ~(G - 1), grp = threadIdx.x / G;
    const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1);
    const int L0 = pl.P - 3;
    const LevelPlan &lz = pl.lv[L0], &la = pl.lv[L0 + 1];
    const int64_t groups = (int64_t)gridDim.x * (256 / G);
    int32_t *zs = s_z[grp], *bs = s_b[grp];
    uint64_t cnt = 0, bytes = 0;
    for (int64_t ri = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G; ri < nrows; ri += groups) {
        const int64_t r = sel ? sel[ri] : ri;
        const int32_t *m = rows + r * L0;
        // walk the shortest of Z's lists, whole (bounds are tested per element: the twins' differ from Z's)
        int wl = 0;
        int64_t wb = g.off[0][m[t3.zl[0]]], we = g.off[0][m[t3.zl[0]] + 1];
        for (int j = 1; j < t3.n_zl; ++j) {
            const int64_t cb = g.off[0][m[t3.zl[j]]], ce = g.off[0][m[t3.zl[j]] + 1];
            if (ce - cb < we - wb) {
                wb = cb;
                we = ce;
                wl = j;
            }
        }
        // neither Z nor the twins take a vertex at or below the smaller of their lower bounds: a long list
        // is entered above it
        if (we - wb > G) {
            int32_t lo_z = -1, lo_b = -1;
            for (int j = 0; j < lz.n_lower; ++j) lo_z = max(lo_z, m[lz.lower_level[j]]);
            for (int j = 0; j < la.n_lower; ++j) lo_b = max(lo_b, m[la.lower_level[j]]);
            const int32_t lo = min(lo_z, lo_b);
            if (lo >= 0) wb = lower_bound_g(g.adj[0], wb, we, lo + 1);
        }
        int nz = 0, nb = 0;
        for (int64_t i0 = wb; i0 < we && nz <= CAP && nb <= CAP; i0 += G) {
            const int64_t i = i0 + sub;
            bool ok = i < we;
            int32_t c = 0;
            if (ok) {
                c = g.adj[0][i];
                for (int t = 0; t < L0; ++t) ok = ok && m[t] != c;
                for (int j = 0; j < t3.n_zl && ok; ++j)
                    if (j != wl) ok = in_list(g, 0, PART_FULL, m[t3.zl[j]], c);
            }
            const bool zok = ok && t3_level_ok(g, lz, m, c);
            bool bok = ok && t3_level_ok(g, la, m, c);
            for (int j = 0; j < t3.n_xl && bok; ++j) bok = in_list(g, 0, PART_FULL, m[t3.xl[j]], c);
            const unsigned long long mz = (__ballot(zok) >> shift) & gmask, mb = (__ballot(bok) >> shift) & gmask;
            const unsigned long long below = (1ull << sub) - 1;
            if (zok) {
                const int p = nz + __popcll(mz & below);
                if (p < CAP) zs[p] = c;
            }
            if (bok) {
                const int p = nb + __popcll(mb & below);
                if (p < CAP) bs[p] = c;
            }
            nz += __popcll(mz);
            nb += __popcll(mb);
        }
        wave_lds_sync();
        if (nz > CAP || nb > CAP) {  // too large for the group: handed back
            if (sub == 0) over[atomicAdd(n_over, 1ull)] = r;
            wave_lds_sync();
            continue;
        }
        if (sub == 0) bytes += level_bytes(g, lz, m, L0);
        for (int zi = sub; zi < nz; zi += G) {
            const int32_t z = zs[zi];
            uint32_t sc = 0;
            int bi = 0;
            if (z >= g.hub_base) {
                // every member of B above the hub base is one bit of z's row of the hub matrix: four
                // independent probes in flight (the loop is bound by their latency, not by instructions)
                const uint32_t *hrow = g.hub + (size_t)(z - g.hub_base) * (size_t)g.hub_wpr;
                for (; bi + 4 <= nb; bi += 4) {
                    int32_t b[4];
                    uint32_t w[4];
                    bool hub[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        b[k] = bs[bi + k];
                        hub[k] = b[k] >= g.hub_base;
                        w[k] = hub[k] ? hrow[(uint32_t)(b[k] - g.hub_base) >> 5] : 0u;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (hub[k]) sc += b[k] != z && ((w[k] >> ((uint32_t)(b[k] - g.hub_base) & 31)) & 1u);
                        else sc += b[k] != z && in_list(g, 0, PART_FULL, z, b[k]);
                    }
                }
            }
            for (; bi < nb; ++bi) {
                const int32_t b = bs[bi];
                sc += b != z && in_list(g, 0, PART_FULL, z, b);
            }
            const uint64_t s_all = sc;
            const uint64_t pairs = s_all * (s_all - (s_all ? 1 : 0));
            cnt += (ti.b_lower_is_a || ti.b_upper_is_a) ? pairs / 2 : pairs;
            bytes += (s_all + 1) * level_bytes(g, la, m, L0 + 1, L0, z) + 4ull * s_all;
        }
        wave_lds_sync();
    }
    cnt = wave_sum_u64(cnt);
    bytes = wave_sum_u64(bytes);
    if (lane == 0) {
        if (cnt) atomicAdd(n_matches, (unsigned long long)cnt);
        if (bytes) atomicAdd(alg_bytes, (unsigned long long)bytes);
    }
}

// Rows `idx[0..n)` of a frontier of k ids per row, gathered into `out`.
__global__ void k_gather_rows(const int32_t *rows, const int64_t *idx, int64_t n, int k, int32_t *out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * k) return;
    out[t] = rows[idx[t / k] * k + t % k];
}

// Last two levels when both are leaf levels hanging off earlier vertices: one
// thread per partial match, candidates counted in closed form: |A| * |B| minus
// the pairs that would bind the same vertex twice.
__global__ void k_leaf_pair(GraphView g, PlanView pl, TailInfo ti, const int32_t *rows, int64_t nrows,
                            unsigned long long *n_matches, unsigned long long *alg_bytes) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int L = pl.P - 2;
    uint64_t cnt = 0, bytes = 0;
    if (r < nrows) {
        const int32_t *m = rows + r * L;
        const LevelPlan &la = pl.lv[L], &lb = pl.lv[L + 1];
        const int32_t sa = m[la.src_level], sb = m[lb.src_level];
        const int64_t ca = leaf_count(g, la, m, L, sa);
        const int64_t cb = ti.pair_same_set ? ca : leaf_count(g, lb, m, L, sb);  // same source, same label: same set
        if (ca > 0 && cb > 0) cnt = (uint64_t)(ca * cb - (ti.pair_same_set ? ca : 0));
        const uint64_t la_len = (uint64_t)(g.off[0][sa + 1] - g.off[0][sa]);
        const uint64_t lb_len = (uint64_t)(g.off[0][sb + 1] - g.off[0][sb]);
        bytes = 4ull * (uint64_t)L + 4ull * la_len;
        if (ca > 0) bytes += (uint64_t)ca * (4ull * (uint64_t)(L + 1) + 4ull * lb_len);
    }
    cnt = wave_sum_u64(cnt);
    bytes = wave_sum_u64(bytes);
    if (lane_id() == 0) {
        if (cnt) atomicAdd(n_matches, (unsigned long long)cnt);
        if (bytes) atomicAdd(alg_bytes, (unsigned long long)bytes);
    }
}

// Sort key of arc p = (v -> w) for the label-grouped copy of the lists:
// (v << 8) | label(w); a stable sort keeps ids ascending inside every label.
__global__ void k_segment_keys(GraphView g, int64_t m, uint64_t *keys, int32_t *vals) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    int32_t lo = 0, hi = g.n;  // last v with off[v] <= p
    while (hi - lo > 1) {
        int32_t mid = lo + ((hi - lo) >> 1);
        if (g.off[0][mid] <= p) lo = mid;
        else hi = mid;
    }
    const int32_t w = g.adj[0][p];
    keys[p] = ((uint64_t)(uint32_t)lo << 8) | g.label[w];
    vals[p] = w;
}

// labseg[v][l] = neighbours of v with a label below l (nlab + 1 entries per vertex).
__global__ void k_label_segments(GraphView g, const uint32_t *labcnt, uint32_t *labseg) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= g.n) return;
    const uint32_t *cnt = labcnt + (size_t)v * (size_t)g.nlab;
    uint32_t *seg = labseg + (size_t)v * (size_t)(g.nlab + 1);
    uint32_t run = 0;
    for (int l = 0; l < g.nlab; ++l) {
        seg[l] = run;
        run += cnt[l];
    }
    seg[g.nlab] = run;
}

// ---------------------------------------------------------------------------
// Leaf suffix: every level from S on is a leaf
// ---------------------------------------------------------------------------
//
// When the last q >= 2 levels are all leaves (one neighbour in the pattern, no
// other requirement than the label), the number of ways to bind them for one
// partial match m of the S inner vertices has a closed form.  Leaves can only
// clash when they carry the same label, so they are grouped by label and,
// inside a label, by the inner vertex they hang off; at most two such groups
// per label are handled (A = free neighbours of s1 with the label, a leaves;
// B = those of s2, b leaves; t = |A ^ B|):
//     W(a, b) = sum_j (-1)^j C(a,j) C(b,j) j! t^(j) (|A|-j)^(a-j) (|B|-j)^(b-j)
// (x^(k) = falling factorial; the sum runs over the numbers j of leaf pairs
// forced onto a common vertex), divided by a! or b! when ordering bounds make
// the leaves of a group interchangeable.  Three groups per label go through the
// general inclusion-exclusion over set partitions (LeafTerm below).  Only t needs a list intersection,
// one per partial match of the inner vertices instead of one partial match per
// leaf binding.  Counts are taken modulo 2^64 like every other count here.
// Algorithmic bytes: the row, the degree lookups and both operands of the
// intersection -- what the closed form needs, not what enumerating would read.

struct LeafClass {
    int32_t label;                 // -1: unlabelled
    int8_t n_grp;                  // 1 to 3 groups
    int8_t src[3], cnt[3], ordered[3], level0[3];  // inner level, leaves, bounds make them a chain, first leaf level
    int16_t term_beg, term_end;    // three groups: this class's terms in the term table (see LeafTerm)
};
// Three groups A1, A2, A3 of one label: the ordered bindings without a clash are
//     sum over the set partitions pi of the leaves of  mu(pi) * prod over blocks B of pi of n[G(B)],
// mu(pi) = prod (-1)^(|B|-1) (|B|-1)!, G(B) = the groups the leaves of B belong to (a bit mask), and
// n[T] = number of free vertices common to the groups of T.  The host adds up the partitions that give
// the same product: a term is a coefficient and the exponent of every n[T], T = 1..7.
struct LeafTerm {
    long long coef;
    uint8_t e[8];
};
struct LeafSuffix {
    int8_t enabled, S, n_cls;
    LeafClass cls[4];
    int8_t cls_of[GPM_MAXP], grp_of[GPM_MAXP];  // class and group of every leaf level
};

// Arithmetic modulo 2^128 (unsigned): sums, differences and products stay exact modulo 2^64 however
// large the true values are; only the divisions by a factorial below need the true value to fit.
typedef unsigned __int128 u128;

__device__ __forceinline__ u128 falling128(long long x, int k) {
    u128 r = 1;
    for (int i = 0; i < k; ++i) r *= (u128)(x - i > 0 ? x - i : 0);
    return r;
}

// Ways to bind a leaves off s1 and b leaves off s2 (see above), ordered tuples.
__device__ __forceinline__ u128 leaf_ways(long long A, long long B, long long t, int a, int b) {
    if (b == 0) return falling128(A, a);
    if (a == 0) return falling128(B, b);
    u128 sum = 0;
    long long binom_a = 1, binom_b = 1, jfact = 1;
    for (int j = 0; j <= a && j <= b; ++j) {
        if (j > 0) {
            binom_a = binom_a * (a - j + 1) / j;
            binom_b = binom_b * (b - j + 1) / j;
            jfact *= j;
        }
        const u128 term = (u128)(binom_a * binom_b * jfact) * falling128(t, j) * falling128(A - j, a - j) *
                          falling128(B - j, b - j);
        if (j & 1) sum -= term;
        else sum += term;
    }
    return sum;
}

// C(x, k) without ever holding more than the result times k: the falling factorial of a hub's degree
// overflows 128 bits from k = 7 on, the binomial itself does not.
__device__ __forceinline__ u128 binom128(long long x, int k) {
    if (x < k) return 0;
    u128 r = 1;
    for (int i = 1; i <= k; ++i) {
        long long num = x - k + i, den = i;  // r = C(x-k+i-1, i-1) so far; r * num / den is an integer
        long long a = num, b = den;
        while (b) {
            const long long t = a % b;
            a = b;
            b = t;
        }
        num /= a;
        den /= a;
        r = r / (u128)den * (u128)num;  // den divides r: it is coprime to num and divides r * num
    }
    return r;
}

__device__ __forceinline__ long long small_factorial(int k) {
    long long r = 1;
    for (int i = 2; i <= k; ++i) r *= i;
    return r;
}

// One wave per partial match of the S inner vertices.
__global__ __launch_bounds__(TAIL_WAVES *GPM_WAVE) void k_leaf_suffix(GraphView g, PlanView pl, LeafSuffix ls,
                                                                      const LeafTerm *terms, const int32_t *rows,
                                                                      int64_t nrows, unsigned long long *n_matches,
                                                                      unsigned long long *alg_bytes) {
    const int wave = threadIdx.x / GPM_WAVE, lane = lane_id();
    const int S = ls.S;
    uint64_t cnt = 0, bytes = 0;
    for (int64_t r = (int64_t)blockIdx.x * TAIL_WAVES + wave; r < nrows; r += (int64_t)gridDim.x * TAIL_WAVES) {
        const int32_t *m = rows + r * S;
        u128 ways = 1;  // meaningful in lane 0
        if (lane == 0) bytes += 4ull * (uint64_t)S;
        for (int c = 0; c < ls.n_cls; ++c) {
            const LeafClass &lc = ls.cls[c];
            const bool segs = lc.label >= 0 && g.adjL;
            const int32_t *arr = segs ? g.adjL : g.adj[0];
            // candidates of group i: the label's segment of its source's list when there is one (ids ascend
            // inside it), else the whole list
            int64_t lb[3] = {0, 0, 0}, le[3] = {0, 0, 0};
            long long n1[3] = {0, 0, 0};  // free neighbours with the label
            for (int i = 0; i < lc.n_grp; ++i) {
                const int32_t si = m[lc.src[i]];
                n1[i] = leaf_count(g, pl.lv[lc.level0[i]], m, S, si);
                if (segs) {
                    const uint32_t *q = g.labseg + (size_t)si * (size_t)(g.nlab + 1) + (size_t)lc.label;
                    lb[i] = g.off[0][si] + q[0], le[i] = g.off[0][si] + q[1];
                } else {
                    lb[i] = g.off[0][si], le[i] = g.off[0][si + 1];
                }
            }
            if (lane == 0) bytes += 20ull * (uint64_t)lc.n_grp;  // two offsets and a label count per group
            // free vertices of list x that are also in list y (and in list z, and in both): lanes over x
            auto common = [&](int x, int y, int z, long long &cxy, long long &cxz, long long &cxyz) {
                uint32_t a = 0, b = 0, ab = 0;
                for (int64_t i = lb[x] + lane; i < le[x]; i += GPM_WAVE) {
                    const int32_t v = arr[i];
                    if (lc.label >= 0 && !segs && g.label[v] != (uint8_t)lc.label) continue;
                    bool free_v = true;
                    for (int t = 0; t < S && free_v; ++t) free_v = m[t] != v;  // bound vertices are not free
                    if (!free_v) continue;
                    const bool iy = contains_g(arr, lb[y], le[y], v);
                    const bool iz = z >= 0 && contains_g(arr, lb[z], le[z], v);
                    a += iy;
                    b += iz;
                    ab += iy && iz;
                }
                cxy = (long long)__shfl(wave_sum_u64(a), 0, GPM_WAVE);
                cxz = (long long)__shfl(wave_sum_u64(b), 0, GPM_WAVE);
                cxyz = (long long)__shfl(wave_sum_u64(ab), 0, GPM_WAVE);
                if (lane == 0)  // merge model: every operand once
                    bytes += 4ull * (uint64_t)((le[x] - lb[x]) + (le[y] - lb[y]) + (z >= 0 ? le[z] - lb[z] : 0));
            };
            u128 w = 0;
            if (lc.n_grp == 1) {
                w = lc.ordered[0] ? binom128(n1[0], lc.cnt[0]) : falling128(n1[0], lc.cnt[0]);
            } else if (lc.n_grp == 2) {
                const int x = le[1] - lb[1] < le[0] - lb[0] ? 1 : 0;
                long long t = 0, unused1, unused2;
                common(x, 1 - x, -1, t, unused1, unused2);
                w = leaf_ways(n1[0], n1[1], t, lc.cnt[0], lc.cnt[1]);
                // the divisions are exact while |A|^a |B|^b stays below 2^128
                if (lc.ordered[0]) w /= (u128)small_factorial(lc.cnt[0]);
                if (lc.ordered[1]) w /= (u128)small_factorial(lc.cnt[1]);
            } else {
                int x = 0;  // the shortest list is walked against the two others, then the shorter of those
                for (int i = 1; i < 3; ++i)
                    if (le[i] - lb[i] < le[x] - lb[x]) x = i;
                int y = (x + 1) % 3, z = (x + 2) % 3;
                if (le[z] - lb[z] < le[y] - lb[y]) {
                    const int t = y;
                    y = z;
                    z = t;
                }
                long long n[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cyz, unused1, unused2;
                common(x, y, z, n[(1 << x) | (1 << y)], n[(1 << x) | (1 << z)], n[7]);
                common(y, z, -1, cyz, unused1, unused2);
                n[(1 << y) | (1 << z)] = cyz;
                n[1] = n1[0], n[2] = n1[1], n[4] = n1[2];
                if (lane == 0) {
                    for (int t = lc.term_beg; t < lc.term_end; ++t) {
                        const LeafTerm tm = terms[t];
                        u128 prod = (u128)(tm.coef < 0 ? -tm.coef : tm.coef);
                        for (int k = 1; k < 8; ++k)
                            for (int e = 0; e < tm.e[k]; ++e) prod *= (u128)n[k];
                        if (tm.coef < 0) w -= prod;
                        else w += prod;
                    }
                    for (int i = 0; i < 3; ++i)
                        if (lc.ordered[i]) w /= (u128)small_factorial(lc.cnt[i]);
                }
            }
            ways *= w;
        }
        if (lane == 0) cnt += (uint64_t)ways;
    }
    cnt = wave_sum_u64(cnt);
    bytes = wave_sum_u64(bytes);
    if (lane == 0) {
        if (cnt) atomicAdd(n_matches, (unsigned long long)cnt);
        if (bytes) atomicAdd(alg_bytes, (unsigned long long)bytes);
    }
}

// Neighbours per label of every vertex (thread per vertex).
__global__ void k_label_counts(GraphView g, uint32_t *labcnt) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= g.n) return;
    uint32_t *row = labcnt + (size_t)v * (size_t)g.nlab;
    for (int64_t i = g.off[0][v]; i < g.off[0][v + 1]; ++i) row[g.label[g.adj[0][i]]] += 1;
}

// ---------------------------------------------------------------------------
// Graph preparation: hub matrix
// ---------------------------------------------------------------------------

// One workgroup per hub vertex: set the bits of its arcs to other hubs.
__global__ void k_hub_fill(GraphView g, uint32_t *bits, uint32_t *summary) {
    const int32_t v = g.hub_base + blockIdx.x;
    if (v >= g.n) return;
    const int64_t e = g.off[0][v + 1];
    int64_t b = lower_bound_g(g.adj[0], g.off[0][v], e, g.hub_base);
    uint32_t *row = bits + (size_t)blockIdx.x * (size_t)g.hub_wpr;
    uint32_t *srow = summary ? summary + (size_t)blockIdx.x * (size_t)g.hub_cwpr : nullptr;
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        uint32_t c = (uint32_t)(g.adj[0][i] - g.hub_base);
        atomicOr(row + (c >> 5), 1u << (c & 31));
        if (srow) atomicOr(srow + (c >> 10), 1u << ((c >> 5) & 31));  // word c >> 5 of the row holds a bit
    }
}

// ---------------------------------------------------------------------------
// Cliques: root-centric bitmap kernel
// ---------------------------------------------------------------------------
//
// For a K-clique under the ordering bounds m0 < m1 < ... every vertex after
// the root lies in U = N+(root), the part of the root's list above the root
// (at most a few thousand ids after the degree renumbering).  One workgroup
// takes one root, stages U in LDS and builds the adjacency of the subgraph
// induced on U as d rows of d bits: row i has bit j set iff U[j] is in
// N+(U[i]) (so j > i).  A row is N+(U[i]) ^ U: for a hub U[i] it is gathered
// from the hub matrix, 64 bits per ballot; otherwise N+(U[i]) is streamed with
// coalesced reads and looked up in the LDS copy of U.  All deeper levels run
// on the bitmaps alone.  The last two levels are fused: the set bits j of the
// current candidate set C are expanded into a list, and groups of lanes each
// take one j, read row j from LDS 16 bytes per lane, AND it with C and add the
// population count -- |C ^ row_j| for 64/L values of j per wave instruction,
// where L = lanes per row.

#define CLQ_MAX_D 2048                 // largest universe (bits per row) of k_clique
#define CLQ_BIG_MAX_D 4096             // ... of the build/count kernel pair (four column tiles)
#ifndef CLQ_GATHERS
#define CLQ_GATHERS 4                  // independent hub-matrix gathers in flight per lane
#endif
#define CLQ_SLOTS 1024                 // result counters are spread over this many slots
#define CLQ_CLASSES 10
#define CLQ_GENERIC 6                  // class of the roots too large for the bitmap kernels
#define CLQ_TINY 7                     // classes 7, 8, 9: universe fits a group of 16, 32, 64 lanes
#define CLQ_TINY_G 16

// small_max: largest universe of the kernels that keep their rows in LDS (256; tests lower it to send
// small universes through the build / count kernel pair).
// tiny_max: largest universe of the lane-group kernel (16, 32 or 64, or 0 when it is not used).
__device__ __forceinline__ int clq_class(int64_t d, int max_d, int small_max, int tiny_max) {
    return d > max_d ? CLQ_GENERIC
         : d <= tiny_max ? (d <= 16 ? CLQ_TINY : d <= 32 ? CLQ_TINY + 1 : CLQ_TINY + 2)
         : d <= min(64, small_max) ? 0 : d <= small_max ? 1 : d <= 512 ? 2 : d <= 1024 ? 3 : d <= 2048 ? 4 : 5;
}

// Sort roots of a chunk into size classes by d = |N+(root)|; each class is a
// compacted list (ballot + prefix popcount, one atomic per wave and class).
//   class 0: K-1 <= d <= 64   1: d <= 256   2: d <= 512   3: d <= 1024
//   4: d <= 2048   5: d <= 4096 (classes 4, 5 only up to max_d)   6: larger (generic path)
//   7, 8, 9: d <= 16, 32, 64 where d <= tiny_max (taken out of classes 0 and 1)
__global__ void k_clique_bin(GraphView g, int K, int max_d, int small_max, int tiny_max, int64_t lo, int64_t hi,
                             int32_t *lists, int64_t list_stride, unsigned long long *counts) {
    int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1;
    int32_t v = 0;
    if (i < hi) {
        v = (int32_t)(g.n - 1 - i);
        int64_t d = g.off[0][v + 1] - g.mid[0][v];
        if (d >= K - 1) cls = clq_class(d, max_d, small_max, tiny_max);
    }
    int lane = lane_id();
    for (int c = 0; c < CLQ_CLASSES; ++c) {
        unsigned long long mask = __ballot(cls == c);
        if (!mask) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(counts + c, (unsigned long long)__popcll(mask));
        base = __shfl(base, 0, GPM_WAVE);
        if (cls == c) lists[c * list_stride + base + __popcll(mask & ((1ull << lane) - 1))] = v;
    }
}

// Work estimate of every root of [lo, hi) (root index i is vertex n-1-i),
// used to cut the root range into chunks of similar cost: the square of the
// list the root's search walks first (N+ under ordering bounds), so a chunk of
// hubs holds few roots and a chunk of leaves holds many.
__global__ void k_root_cost(GraphView g, int oriented, int min_d, int64_t lo, int64_t hi, int64_t *cost) {
    int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    int32_t v = (int32_t)(g.n - 1 - i);
    int64_t d = g.off[0][v + 1] - (oriented ? g.mid[0][v] : g.off[0][v]);
    cost[i - lo] = d < min_d ? 1 : d * d + 8;
}

// Size-class histogram of the roots of every chunk (the classes of k_clique_bin).
__global__ void k_chunk_classes(GraphView g, int K, int max_d, int small_max, int tiny_max, int64_t lo, int64_t hi,
                                const int64_t *bounds, int nchunks, unsigned int *table) {
    int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    int32_t v = (int32_t)(g.n - 1 - i);
    int64_t d = g.off[0][v + 1] - g.mid[0][v];
    if (d < K - 1) return;
    int cls = clq_class(d, max_d, small_max, tiny_max);
    int a = 0, b = nchunks;  // last chunk whose first root is <= i
    while (b - a > 1) {
        int m = (a + b) >> 1;
        if (bounds[m] <= i) a = m;
        else b = m;
    }
    atomicAdd(table + a * CLQ_CLASSES + cls, 1u);
}

struct u32x4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ u32x4 and4(u32x4 a, u32x4 b) { return {a.x & b.x, a.y & b.y, a.z & b.z, a.w & b.w}; }
__device__ __forceinline__ uint32_t popc4(u32x4 a) { return __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w); }
__device__ __forceinline__ u32x4 load4(const uint32_t *p) {
    uint4 v = *reinterpret_cast<const uint4 *>(p);
    return {v.x, v.y, v.z, v.w};
}

// popc(x) + acc in one instruction (the compiler otherwise counts into a fresh register and adds).
__device__ __forceinline__ uint32_t popc_add(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t and_popc_add4(u32x4 c, u32x4 r, uint32_t acc) {
    acc = popc_add(c.x & r.x, acc);
    acc = popc_add(c.y & r.y, acc);
    acc = popc_add(c.z & r.z, acc);
    return popc_add(c.w & r.w, acc);
}

// Open-addressing hash of the universe U in LDS, used while the rows are built:
// slot = (id & ~0x1fff) | (position + 1), 0 = empty.  The 19 tag bits reject
// almost every foreign id without touching U.
__device__ __forceinline__ uint32_t clq_hash(int32_t id, int shift) { return ((uint32_t)id * 2654435761u) >> shift; }

__device__ __forceinline__ void clq_hash_insert(uint32_t *tab, uint32_t mask, int shift, int32_t id, int pos) {
    uint32_t s = clq_hash(id, shift);
    const uint32_t val = ((uint32_t)id & ~0x1fffu) | (uint32_t)(pos + 1);
    while (atomicCAS(tab + s, 0u, val) != 0u) s = (s + 1) & mask;
}

// Position of id in U, or -1.
__device__ __forceinline__ int clq_hash_find(const uint32_t *tab, uint32_t mask, int shift, const int32_t *U,
                                             int32_t id) {
    uint32_t s = clq_hash(id, shift);
    for (;;) {
        uint32_t x = tab[s];
        if (x == 0u) return -1;
        if (((x ^ (uint32_t)id) & ~0x1fffu) == 0u) {
            int pos = (int)(x & 0x1fffu) - 1;
            if (U[pos] == id) return pos;
        }
        s = (s + 1) & mask;
    }
}

// Geometry of one root's bitmaps.
struct ClqGeom {
    int d;    // universe size
    int Wp;   // 32-bit words per row, a multiple of 4 (rows are 16-byte aligned)
    int L;    // lanes that hold one row, 4 words each: power of two, 1..16
    int lg;   // log2(L)
};

// Per-wave scratch in LDS.
struct ClqWave {
    uint16_t *J;        // expanded list of set-bit positions
    uint32_t *cbuf;     // one candidate set per nesting depth, Wp words each
};

// Expand the set bits of the Wp-word bitmap at `words` (LDS) into J; returns
// how many.  Lane l looks at bit l&31 of word 2t + (l>>5); one ballot places
// 64 bits at a time.
__device__ __forceinline__ int clq_expand(const uint32_t *words, int Wp, uint16_t *J, int lane) {
    int cnt = 0;
    for (int w0 = 0; w0 < Wp; w0 += 2) {
        uint32_t word = words[w0 + (lane >> 5)];
        bool bit = (word >> (lane & 31)) & 1u;
        unsigned long long mask = __ballot(bit);
        if (bit) J[cnt + __popcll(mask & ((1ull << lane) - 1))] = (uint16_t)((w0 << 5) + lane);
        cnt += __popcll(mask);
    }
    return cnt;
}

// Same for a bitmap held in registers, lane l holding word l (Wp <= 64).
__device__ __forceinline__ int clq_expand_reg(uint32_t mine, int Wp, uint16_t *J, int lane) {
    int cnt = 0;
    for (int w0 = 0; w0 < Wp; w0 += 2) {
        uint32_t lo = __builtin_amdgcn_readlane(mine, w0), hi = __builtin_amdgcn_readlane(mine, w0 + 1);
        if ((lo | hi) == 0) continue;  // same for the whole wave
        uint32_t word = lane < 32 ? lo : hi;
        bool bit = (word >> (lane & 31)) & 1u;
        unsigned long long mask = __ballot(bit);
        if (bit) J[cnt + __popcll(mask & ((1ull << lane) - 1))] = (uint16_t)((w0 << 5) + lane);
        cnt += __popcll(mask);
    }
    return cnt;
}

// Ways to complete a clique with R more vertices out of the candidate set C.
// C is held 4 words per lane by lanes 0..L-1 and replicated in every group of
// L lanes.  Returns a per-lane partial sum (each completed clique is counted
// in exactly one lane).  `depth` indexes the per-depth LDS copies of C.
template <int R>
__device__ __forceinline__ uint64_t clq_count(u32x4 C, const uint32_t *c_words, const uint32_t *rows,
                                              const ClqGeom &q, const ClqWave &ws, int depth, int lane,
                                              uint64_t &nodes) {
    const int sub = lane & (q.L - 1);
    const bool lane_ok = 4 * sub < q.Wp;  // L is Wp/4 rounded up to a power of two
    if constexpr (R == 1) {
        return lane < q.L ? popc4(C) : 0u;
    } else if constexpr (R == 2) {
        // fused last two levels: |C ^ row_j| for every j in C, 64/L at a time
        wave_lds_sync();
        const int cnt = clq_expand(c_words, q.Wp, ws.J, lane);
        wave_lds_sync();
        if (depth >= 0) nodes += (uint64_t)cnt;
        uint32_t acc = 0;
        const int G = GPM_WAVE >> q.lg;
        if (lane_ok) {
            const uint32_t *base = rows + 4 * sub;
            // row j starts j * Wp words in: a shift when Wp is a power of two (always, up to 256 vertices)
            auto run = [&](auto row_of) {
                int t = lane >> q.lg;
                for (; t + 3 * G < cnt; t += 4 * G) {  // four independent row reads in flight
                    const u32x4 r0 = load4(row_of(ws.J[t]));
                    const u32x4 r1 = load4(row_of(ws.J[t + G]));
                    const u32x4 r2 = load4(row_of(ws.J[t + 2 * G]));
                    const u32x4 r3 = load4(row_of(ws.J[t + 3 * G]));
                    acc = and_popc_add4(C, r0, acc);
                    acc = and_popc_add4(C, r1, acc);
                    acc = and_popc_add4(C, r2, acc);
                    acc = and_popc_add4(C, r3, acc);
                }
                for (; t < cnt; t += G) acc = and_popc_add4(C, load4(row_of(ws.J[t])), acc);
            };
            if ((q.Wp & (q.Wp - 1)) == 0) {
                const int sh = q.lg + 2;  // Wp = 4 L
                run([&](uint32_t j) { return base + (j << sh); });
            } else {
                run([&](uint32_t j) { return base + (size_t)j * q.Wp; });
            }
        }
        return acc;
    } else {
        uint64_t acc = 0;
        uint32_t *next = ws.cbuf + (size_t)(depth + 1) * q.Wp;
        for (int w = 0; w < q.Wp; ++w) {
            uint32_t word = __builtin_amdgcn_readfirstlane(c_words[w]);
            while (word) {
                int j = (w << 5) + __builtin_ctz(word);
                word &= word - 1;
                u32x4 C2 = {0u, 0u, 0u, 0u};
                if (lane_ok) C2 = and4(C, load4(rows + (size_t)j * q.Wp + 4 * sub));
                wave_lds_sync();
                if (lane < q.L && lane_ok)
                    *reinterpret_cast<uint4 *>(next + 4 * sub) = make_uint4(C2.x, C2.y, C2.z, C2.w);
                wave_lds_sync();
                if (depth >= 0) nodes += 1;
                acc += clq_count<R - 1>(C2, next, rows, q, ws, depth + 1, lane, nodes);
            }
        }
        return acc;
    }
}

// One workgroup (BLOCK threads) per root.  Dynamic LDS layout, 32-bit words:
//   U[dcap] | Dp[dcap] | work[max(NW * (dcap/2 + DEPTHS*wcap), 2*dcap)] | rows[dcap * wcap] (IN_LDS only)
// `work` holds the hash of U while the rows are built and afterwards, per
// wave, the expanded list J and the nested candidate sets.  With !IN_LDS the
// rows live in this workgroup's slab of `scratch`.
template <int K, int BLOCK, bool IN_LDS, bool STATS>
__global__ __launch_bounds__(BLOCK) void k_clique(GraphView g, const int32_t *roots, int64_t nroots, int dcap,
                                                  int probe_bias, uint32_t *scratch,
                                                  unsigned long long *n_matches, unsigned long long *alg_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int NW = BLOCK / GPM_WAVE;
    constexpr int DEPTHS = K > 3 ? K - 3 : 1;  // nested candidate sets kept in LDS
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / GPM_WAVE;
    const int wcap = dcap / 32 < 4 ? 4 : dcap / 32;
    int32_t *U = (int32_t *)smem;
    uint32_t *Dp = smem + dcap;
    uint32_t *work = smem + 2 * dcap;
    const int per_wave = dcap / 2 + DEPTHS * wcap;
    const int work_words = NW * per_wave > 2 * dcap ? NW * per_wave : 2 * dcap;
    uint32_t *wave_base = work + (size_t)wave * (size_t)per_wave;
    ClqWave ws;
    ws.J = (uint16_t *)wave_base;
    ws.cbuf = wave_base + dcap / 2;
    uint32_t *htab = work;
    uint32_t *rows = IN_LDS ? work + work_words
                            : scratch + (size_t)blockIdx.x * ((size_t)CLQ_MAX_D * (CLQ_MAX_D / 32));
    const int64_t *off = g.off[0], *mid = g.mid[0];
    const int32_t *adj = g.adj[0];
    uint64_t total = 0, bytes = 0;

    for (int64_t ri = blockIdx.x; ri < nroots; ri += gridDim.x) {
        const int32_t u = roots[ri];
        const int64_t ub = mid[u];
        ClqGeom q;
        q.d = (int)(off[u + 1] - ub);
        q.Wp = (((q.d + 31) >> 5) + 3) & ~3;
        q.lg = q.Wp <= 4 ? 0 : q.Wp <= 8 ? 1 : q.Wp <= 16 ? 2 : q.Wp <= 32 ? 3 : 4;
        q.L = 1 << q.lg;
        const int d = q.d, Wp = q.Wp;
        int hbits = 1;  // table of 2^hbits >= 2*d slots
        while ((1 << hbits) < 2 * d) ++hbits;
        const uint32_t hmask = (1u << hbits) - 1;
        const int hshift = 32 - hbits;
        // Lists ascend, so a universe whose first vertex is a hub holds hubs only: every row is then
        // gathered from the hub matrix and neither the hash of U nor the list lengths are needed.
        const bool hubs_only = probe_bias == 0 && d > 0 && adj[ub] >= g.hub_base;
        const bool need_dp = STATS || probe_bias != 0;
        if (!hubs_only) {
            for (int i = tid; i <= (int)hmask; i += BLOCK) htab[i] = 0;
            __syncthreads();
        }
        for (int i = tid; i < d; i += BLOCK) {
            int32_t v = adj[ub + i];
            U[i] = v;
            if (need_dp) Dp[i] = (uint32_t)(off[v + 1] - mid[v]);
            if (!hubs_only) clq_hash_insert(htab, hmask, hshift, v, i);
        }
        __syncthreads();

        // ---- build the rows: row i = N+(U[i]) ^ U, as bits over positions in U
        uint64_t arcs_bytes = 0;  // per lane: sum over its arcs i->j of 12 + 4*(d + Dp[i] + Dp[j])
        // The hub rows of a universe of at most 256 vertices: a row has positions to test only in the blocks
        // of 64 from the one that holds i + 1 on.  The (row, block)
        // pairs that hold something are therefore numbered block by block -- block b is tested by the rows
        // 0 .. min(d, 64 b + 63) - 1 -- and a wave takes four of them per trip, whichever rows they belong
        // to: four useful gathers in flight per trip instead of one to four (one for most rows of a small
        // universe).
        // U ascends, so its hubs are its last vertices: rows h0 .. d - 1 are packed, the rows before them walk
        // their lists below.
        const bool packed = probe_bias == 0 && g.hub != nullptr && d <= CLQ_GATHERS * GPM_WAVE;
        int h0 = d;  // first position of U that holds a hub
        if (packed) {
            int lo = 0;
            while (lo < h0) {
                const int m = (lo + h0) >> 1;
                if (U[m] < g.hub_base) lo = m + 1;
                else h0 = m;
            }
        }
        if (packed && h0 < d) {
            const int nb = (d + GPM_WAVE - 1) / GPM_WAVE;
            int first[CLQ_GATHERS + 1];  // number of the first pair of block b
            first[0] = 0;
#pragma unroll
            for (int bk = 0; bk < CLQ_GATHERS; ++bk)
                first[bk + 1] = first[bk] + (bk < nb ? max(0, min(d, GPM_WAVE * bk + GPM_WAVE - 1) - h0) : 0);
            const int pairs = first[CLQ_GATHERS];
            // the words no pair writes are zeros: blocks below the one that holds i + 1, blocks past d
            const int wsh = Wp == 4 ? 2 : 3;  // Wp is 4 or 8 up to 256 vertices
            for (int x = tid + (h0 << wsh); x < (d << wsh); x += BLOCK) {
                const int i = x >> wsh, blk = (x & (Wp - 1)) >> 1;
                if (blk < ((i + 1) >> 6) || blk >= nb) rows[x] = 0;
            }
            for (int t0 = CLQ_GATHERS * wave; t0 < pairs; t0 += CLQ_GATHERS * NW) {
                uint32_t word[CLQ_GATHERS], sh[CLQ_GATHERS], col[CLQ_GATHERS], sum[CLQ_GATHERS];
                bool live[CLQ_GATHERS];
                int row_of[CLQ_GATHERS], blk_of[CLQ_GATHERS];
#pragma unroll
                for (int k = 0; k < CLQ_GATHERS; ++k) {
                    const int t = min(t0 + k, max(pairs, 1) - 1);  // a trip past the end repeats the last pair, and drops it below
                    const int bk = (t >= first[1]) + (t >= first[2]) + (t >= first[3]);
                    const int i = h0 + t - first[bk];
                    row_of[k] = i;
                    blk_of[k] = bk;
                    const int j = bk * GPM_WAVE + lane;
                    live[k] = t0 + k < pairs && j > i && j < d;
                    col[k] = live[k] ? (uint32_t)(U[j] - g.hub_base) : 0u;
                    sum[k] = 0;
                    word[k] = 0;
                    sh[k] = col[k] & 31;
                    if (live[k]) {
                        const size_t hr = (size_t)(U[i] - g.hub_base);
                        if (col[k] >= (uint32_t)g.hub_dense || !g.hubc) word[k] = g.hub[hr * (size_t)g.hub_wpr + (col[k] >> 5)];
                        else sum[k] = g.hubc[hr * (size_t)g.hub_cwpr + (col[k] >> 10)];
                    }
                }
#pragma unroll
                for (int k = 0; k < CLQ_GATHERS; ++k)
                    if ((sum[k] >> ((col[k] >> 5) & 31)) & 1u)
                        word[k] = g.hub[(size_t)(U[row_of[k]] - g.hub_base) * (size_t)g.hub_wpr + (col[k] >> 5)];
#pragma unroll
                for (int k = 0; k < CLQ_GATHERS; ++k) {
                    if (t0 + k >= pairs) break;
                    const bool bit = live[k] && ((word[k] >> sh[k]) & 1u);
                    unsigned long long mask = __ballot(bit);
                    if (STATS && bit)
                        arcs_bytes += 12ull + 4ull * ((uint64_t)d + Dp[row_of[k]]) + 4ull * Dp[blk_of[k] * GPM_WAVE + lane];
                    if (lane == 0) {
                        uint32_t *row = rows + ((size_t)row_of[k] << wsh) + 2 * blk_of[k];
                        row[0] = (uint32_t)mask;
                        row[1] = (uint32_t)(mask >> 32);
                    }
                }
            }
        }
        for (int i = wave; i < h0; i += NW) {
            const int32_t v = U[i];
            uint32_t *row = rows + (size_t)i * Wp;
            const uint64_t per_arc = STATS ? 12ull + 4ull * ((uint64_t)d + Dp[i]) : 0ull;
            // a hub's row is gathered from the hub matrix when its list is long
            // next to the number of positions to test (probe_bias/16 is the ratio)
            if (v >= g.hub_base &&
                (probe_bias == 0 || (int64_t)Dp[i] * 16 >= (int64_t)probe_bias * (int64_t)(d - 1 - i))) {
                const uint32_t *hrow = g.hub + (size_t)(v - g.hub_base) * (size_t)g.hub_wpr;
                const uint32_t *crow = g.hubc ? g.hubc + (size_t)(v - g.hub_base) * (size_t)g.hub_cwpr : nullptr;
                // 256 positions per trip: four independent gathers in flight (Wp*32 is a multiple of 128);
                // the summary of the row is asked first, the matrix only for words that hold a bit
                for (int j0 = 0; j0 < Wp * 32; j0 += CLQ_GATHERS * GPM_WAVE) {
                    uint32_t word[CLQ_GATHERS];
                    uint32_t sh[CLQ_GATHERS];
                    bool live[CLQ_GATHERS];
                    uint32_t col[CLQ_GATHERS], sum[CLQ_GATHERS];
#pragma unroll
                    for (int k = 0; k < CLQ_GATHERS; ++k) {
                        const int j = j0 + k * GPM_WAVE + lane;
                        live[k] = j > i && j < d;
                        col[k] = live[k] ? (uint32_t)(U[j] - g.hub_base) : 0u;
                        sum[k] = 0;
                        word[k] = 0;
                        sh[k] = col[k] & 31;
                        if (live[k]) {
                            if (col[k] >= (uint32_t)g.hub_dense) word[k] = hrow[col[k] >> 5];
                            else sum[k] = crow[col[k] >> 10];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < CLQ_GATHERS; ++k)
                        if ((sum[k] >> ((col[k] >> 5) & 31)) & 1u) word[k] = hrow[col[k] >> 5];
#pragma unroll
                    for (int k = 0; k < CLQ_GATHERS; ++k) {
                        const int jb = j0 + k * GPM_WAVE;
                        if (jb >= Wp * 32) break;
                        const bool bit = live[k] && ((word[k] >> sh[k]) & 1u);
                        unsigned long long mask = __ballot(bit);
                        if (STATS && bit) arcs_bytes += per_arc + 4ull * Dp[jb + lane];
                        if (lane == 0) {
                            row[jb >> 5] = (uint32_t)mask;
                            row[(jb >> 5) + 1] = (uint32_t)(mask >> 32);
                        }
                    }
                }
            } else {
                for (int w = lane; w < Wp; w += GPM_WAVE) row[w] = 0;
                if (!IN_LDS) __threadfence_block();
                wave_lds_sync();
                const int64_t vb = mid[v], ve = off[v + 1];
                for (int64_t p0 = vb; p0 < ve; p0 += 4 * GPM_WAVE) {  // four list reads in flight per lane
                    int32_t w[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int64_t p = p0 + k * GPM_WAVE + lane;
                        w[k] = p < ve ? adj[p] : -1;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (w[k] < 0) continue;
                        int j = clq_hash_find(htab, hmask, hshift, U, w[k]);
                        if (j >= 0) {
                            atomicOr(row + (j >> 5), 1u << (j & 31));
                            if (STATS) arcs_bytes += per_arc + 4ull * Dp[j];
                        }
                    }
                }
            }
        }
        if (!IN_LDS) {  // rows were written with stores and L2 atomics: drop this CU's stale L1 lines
            __threadfence();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();

        // ---- count on the bitmaps
        uint64_t nodes = 0;
        const int sub = lane & (q.L - 1);
        for (int i = wave; i < d; i += NW) {
            const uint32_t *row = rows + (size_t)i * Wp;
            u32x4 C = {0u, 0u, 0u, 0u};
            if (4 * sub < Wp) C = load4(row + 4 * sub);
            total += clq_count<K - 2>(C, row, rows, q, ws, -1, lane, nodes);
        }

        // merge-model bytes of the levels fused here:
        //   root -> level 1        : 4*1 + 4*d
        //   (root,i) -> level 2    : 4*2 + 4*(d + Dp[i])                  for every i
        //   (root,i,j) -> level 3  : 4*3 + 4*(d + Dp[i] + Dp[j])         for every arc i->j (K >= 4)
        //   deeper internal nodes (K >= 5) at 4*4 + 4*d each: a lower bound
        if (STATS) {
            uint64_t b = K >= 4 ? arcs_bytes : 0;
            for (int i = tid; i < d; i += BLOCK) b += 8ull + 4ull * ((uint64_t)d + Dp[i]);
            if (tid == 0) b += 4ull + 4ull * (uint64_t)d;
            if (K >= 5 && lane == 0) b += nodes * (16ull + 4ull * (uint64_t)d);
            bytes += b;
        }
        __syncthreads();
    }
    total = wave_sum_u64(total);
    bytes = wave_sum_u64(bytes);
    if (lane == 0) {
        if (total) atomicAdd(n_matches + (blockIdx.x & (CLQ_SLOTS - 1)), (unsigned long long)total);
        if (bytes) atomicAdd(alg_bytes + (blockIdx.x & (CLQ_SLOTS - 1)), (unsigned long long)bytes);
    }
}

// Triangles / 4-cliques of roots whose universe has at most G = 16, 32 or 64 vertices -- most roots of
// a power-law graph.  A wave per root with its rows in LDS would leave most lanes idle and spend its
// time on per-root set-up; here a group of G lanes takes a root and everything stays in registers:
// lane i holds U[i] and builds row i = the positions j > i with U[i] -> U[j] (one hub-matrix bit, or one
// binary search in N+(U[i]) per pair, four pairs in flight; the other vertex is fetched from its lane),
// then adds |row_i ^ row_j| over the set bits j of its row, rows again fetched from their lanes.
// grid = ceil(nroots / (256 / G)) workgroups of 256 threads.
// STATS: also the merge-model bytes of the three levels, as k_clique sums them.
template <int G>
struct TinyRow { typedef uint32_t type; };
template <>
struct TinyRow<64> { typedef uint64_t type; };

template <int K, int G, bool STATS>
__global__ __launch_bounds__(256) void k_clique_tiny(GraphView g, const int32_t *roots, int64_t nroots,
                                                     unsigned long long *n_matches, unsigned long long *alg_bytes) {
    static_assert(K == 3 || K == 4, "lane-group kernel exists for triangles and 4-cliques");
    typedef typename TinyRow<G>::type row_t;
    const int lane = lane_id(), sub = lane & (G - 1), base = lane & ~(G - 1);
    const int64_t ri = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t *off = g.off[0], *mid = g.mid[0];
    const int32_t *adj = g.adj[0];
    int d = 0;
    int32_t my_v = -1;
    int64_t my_b = 0, my_e = 0;
    if (ri < nroots) {
        const int32_t u = roots[ri];
        const int64_t ub = mid[u];
        d = (int)(off[u + 1] - ub);
        if (sub < d) {
            my_v = adj[ub + sub];
            my_b = mid[my_v];
            my_e = off[my_v + 1];
        }
    }
    int dmax = d;  // the largest universe among the groups of this wave bounds the loops
    if (G <= 32) dmax = max(dmax, __shfl_xor(dmax, 32, GPM_WAVE));
    if (G <= 16) dmax = max(dmax, __shfl_xor(dmax, 16, GPM_WAVE));
    const bool my_hub = my_v >= g.hub_base;  // then every vertex above it is a hub as well
    const uint32_t *hrow = my_hub ? g.hub + (size_t)(my_v - g.hub_base) * (size_t)g.hub_wpr : nullptr;
    const uint32_t *crow = my_hub && g.hubc ? g.hubc + (size_t)(my_v - g.hub_base) * (size_t)g.hub_cwpr : nullptr;
    const uint32_t my_dp = (uint32_t)(my_e - my_b);
    const uint64_t per_arc = 12ull + 4ull * ((uint64_t)d + my_dp);
    uint64_t bytes = 0;
    if (STATS && sub < d) bytes = 8ull + 4ull * ((uint64_t)d + my_dp) + (sub == 0 ? 4ull + 4ull * (uint64_t)d : 0ull);
    row_t row = 0;
    for (int j0 = 1; j0 < dmax; j0 += 4) {  // four probes in flight per lane
        int32_t w[4];
        uint32_t word[4], sum[4], dpw[4];
        bool live[4], in[4];
        int64_t sb[4], se[4];  // binary searches of the lanes whose vertex is not a hub, advanced together
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + k;
            w[k] = __shfl(my_v, base + (j & (G - 1)), GPM_WAVE);
            dpw[k] = STATS ? (uint32_t)__shfl((int)my_dp, base + (j & (G - 1)), GPM_WAVE) : 0u;
            live[k] = j < G && j > sub && j < d;
            word[k] = 0;
            sum[k] = 0;
            if (live[k] && my_hub) {  // dense columns straight from the matrix, the others ask the summary first
                const uint32_t c = (uint32_t)(w[k] - g.hub_base);
                if (c >= (uint32_t)g.hub_dense) word[k] = hrow[c >> 5];
                else sum[k] = crow[c >> 10];
            }
            const bool srch = live[k] && !my_hub;
            sb[k] = srch ? my_b : 0;
            se[k] = srch ? my_e : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // the matrix itself only for the words that hold a bit
            const uint32_t c = (uint32_t)(w[k] - g.hub_base);
            if ((sum[k] >> ((c >> 5) & 31)) & 1u) word[k] = hrow[c >> 5];
        }
        while ((sb[0] < se[0]) | (sb[1] < se[1]) | (sb[2] < se[2]) | (sb[3] < se[3])) {
            int32_t x[4];
            int64_t m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[k] = sb[k] + ((se[k] - sb[k]) >> 1);
                x[k] = sb[k] < se[k] ? adj[m[k]] : 0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (sb[k] < se[k]) {
                    if (x[k] < w[k]) sb[k] = m[k] + 1;
                    else se[k] = m[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            in[k] = false;
            if (!live[k]) continue;
            // a search ends on the first entry >= w[k], if there is one
            in[k] = my_hub ? (word[k] >> ((uint32_t)(w[k] - g.hub_base) & 31)) & 1u
                           : (sb[k] < my_e && adj[sb[k]] == w[k]);
            row |= (row_t)in[k] << (j0 + k);
            if (STATS && K >= 4 && in[k]) bytes += per_arc + 4ull * dpw[k];
        }
    }
    uint32_t acc = 0;
    if (K == 3) {
        acc = G == 64 ? (uint32_t)__popcll((uint64_t)row) : (uint32_t)__popc((uint32_t)row);
    } else {
        for (int j = 1; j < dmax; ++j) {
            row_t rj;
            if (G == 64) {
                const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)row, base + j, GPM_WAVE);
                const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)((uint64_t)row >> 32), base + j, GPM_WAVE);
                rj = (row_t)(((uint64_t)hi << 32) | lo);
            } else {
                rj = (row_t)(uint32_t)__shfl((int)(uint32_t)row, base + (j & (G - 1)), GPM_WAVE);
            }
            if ((row >> j) & 1u) acc += G == 64 ? (uint32_t)__popcll((uint64_t)(row & rj)) : (uint32_t)__popc((uint32_t)(row & rj));
        }
    }
    const uint64_t total = wave_sum_u64((uint64_t)acc);
    if (STATS) bytes = wave_sum_u64(bytes);
    if (lane == 0) {
        if (total) atomicAdd(n_matches + (blockIdx.x & (CLQ_SLOTS - 1)), (unsigned long long)total);
        if (STATS && bytes) atomicAdd(alg_bytes + (blockIdx.x & (CLQ_SLOTS - 1)), (unsigned long long)bytes);
    }
}

__global__ void k_sum_slots(const unsigned long long *slots, int n, unsigned long long *out) {
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += slots[i];
    s = wave_sum_u64(s);
    if (lane_id() == 0 && s) atomicAdd(out, s);
}

// ---------------------------------------------------------------------------
// 4-cliques of large universes (256 < d <= 2048): rows built into HBM by
// many small workgroups, then counted from LDS tiles
// ---------------------------------------------------------------------------
//
// A root with a large N+ is too much for one workgroup: its d rows are built
// by d/64 workgroups at full occupancy (they only need U and its hash in LDS)
// and written to a per-root region of a rows buffer in HBM; the count then
// runs as (root, tile pair, segment) tasks.  Columns are cut into tiles of
// 1024 bits.  A task holds the rows j of one row tile, restricted to one
// column tile, in LDS (at most 128 KiB), walks a segment of the rows i, and
// adds |row_i ^ row_j| over that column tile for every j of the row tile set
// in row_i.  With two tiles A (first 1024 positions) and B the pairs are
// (A,A), (A,B) and (B,B): every 4-clique (root, i < j < l) falls in exactly
// one of them by where j and l lie.
//
// Rows buffer, per root (32-bit words): d <= 1024: rows[i][W0], W0 = words of
// d bits rounded up to 4.  d > 1024: up to four column tiles of 1024 bits,
// tile t stored as rows[i][W_t] for i < min(d, 1024 (t + 1)) (see big_off).
// The count runs over the tile pairs (a, b), a <= b: j in tile a, l in tile b.

// Column tiles of one root's rows: tile t covers columns [1024 t, min(d, 1024 (t + 1))) and holds the
// rows that can have bits there, i < min(d, 1024 (t + 1)); every tile but the last is 1024 x 32 words
// per 1024 rows, so tile t starts 32768 * t (t + 1) / 2 words into the root's region.
__device__ __forceinline__ int big_tiles(int d) { return d <= 1024 ? 1 : (d + 1023) >> 10; }
__device__ __forceinline__ int big_rows(int d, int t) { return min(d, 1024 * (t + 1)); }
__device__ __forceinline__ int big_words(int d, int t) {
    return (((big_rows(d, t) - 1024 * t + 31) >> 5) + 3) & ~3;
}
__device__ __forceinline__ int64_t big_off(int t) { return (int64_t)32768 * (t * (t + 1) / 2); }

#define BIG_ROWS_PER_WG 64

// grid = nroots * (dcap / 64) workgroups of 256 threads; workgroup (ri, rb)
// builds rows [64*rb, 64*rb + 64) of root ri.  K == 3 only counts the bits
// (triangles).
//
// ALL_HUB: every vertex of every universe of the launch is a hub (the host
// checks the lowest root of the chunk), so every row is gathered from the hub
// matrix and nothing but U is staged: LDS (words) U[dcap] | stage[4][128].
// Otherwise rows of non-hubs are hashed list walks: U[dcap] | Dp[dcap] |
// hash[2*dcap] | stage[4][128].
// STATS: also sum the merge-model bytes of the levels this kernel and the count
// kernel do (needs Dp, and an add per set bit in the inner loops); off in timed
// runs, on in the accounting search.
template <int K, bool STATS, bool ALL_HUB>
__global__ __launch_bounds__(256) void k_clq_build(GraphView g, const int32_t *roots, int64_t nroots, int dcap,
                                                   int probe_bias, uint32_t *rows_buf, int64_t stride_words,
                                                   unsigned long long *n_matches, unsigned long long *alg_bytes,
                                                   unsigned long long *alg_bytes3) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / GPM_WAVE;
    const int RB = dcap / BIG_ROWS_PER_WG;
    const int64_t ri = blockIdx.x / RB;
    const int i0 = (int)(blockIdx.x % RB) * BIG_ROWS_PER_WG;
    if (ri >= nroots) return;
    const int64_t *off = g.off[0], *mid = g.mid[0];
    const int32_t *adj = g.adj[0];
    const int32_t u = roots[ri];
    const int64_t ub = mid[u];
    const int d = (int)(off[u + 1] - ub);
    if (i0 >= d) return;
    constexpr bool NEED_DP = STATS || !ALL_HUB;  // Dp also steers the gather / list-walk choice of a hub row
    int32_t *U = (int32_t *)smem;
    uint32_t *Dp = smem + dcap;
    uint32_t *htab = smem + 2 * dcap;
    uint32_t *stage = smem + (ALL_HUB && !STATS ? dcap : 4 * dcap) + wave * 128;
    const int T = big_tiles(d);
    uint32_t *region = rows_buf + ri * stride_words;

    uint32_t hmask = 0;
    int hshift = 0;
    if (!ALL_HUB) {
        int hbits = 1;
        while ((1 << hbits) < 2 * d) ++hbits;
        hmask = (1u << hbits) - 1;
        hshift = 32 - hbits;
        for (int i = tid; i <= (int)hmask; i += 256) htab[i] = 0;
        __syncthreads();
    }
    for (int i = tid; i < d; i += 256) {
        int32_t v = adj[ub + i];
        U[i] = v;
        if (NEED_DP) Dp[i] = (uint32_t)(off[v + 1] - mid[v]);
        if (!ALL_HUB) clq_hash_insert(htab, hmask, hshift, v, i);
    }
    __syncthreads();

    const int Wall = 32 * (T - 1) + big_words(d, T - 1);  // words of a whole row in the stage
    uint64_t bytes = 0, bytes3 = 0, tri = 0;  // bytes3: level 3, the work of the count kernel on these rows
    const int i_end = min(d, i0 + BIG_ROWS_PER_WG);
    for (int i = i0 + wave; i < i_end; i += 4) {
        const int32_t v = U[i];
        const uint64_t per_arc = STATS ? 12ull + 4ull * ((uint64_t)d + Dp[i]) : 0ull;
        bool gather = ALL_HUB;
        if (!ALL_HUB)
            gather = v >= g.hub_base && (int64_t)Dp[i] * 16 >= (int64_t)probe_bias * (int64_t)(d - 1 - i);
        if (gather) {
            const uint32_t *hrow = g.hub + (size_t)(v - g.hub_base) * (size_t)g.hub_wpr;
            // positions up to i hold no bits: whole trips below i are zeros, no loads
            const int jfirst = ((i + 1) / (CLQ_GATHERS * GPM_WAVE)) * (CLQ_GATHERS * GPM_WAVE);
            if (K != 3)
                for (int w = lane; w < (jfirst >> 5); w += GPM_WAVE) stage[w] = 0;
            for (int j0 = jfirst; j0 < Wall * 32; j0 += CLQ_GATHERS * GPM_WAVE) {
                uint32_t word[CLQ_GATHERS], sh[CLQ_GATHERS];
                bool live[CLQ_GATHERS];
#pragma unroll
                for (int k = 0; k < CLQ_GATHERS; ++k) {
                    const int j = j0 + k * GPM_WAVE + lane;
                    live[k] = j > i && j < d;
                    word[k] = 0;
                    sh[k] = 0;
                    if (live[k]) {
                        uint32_t c = (uint32_t)(U[j] - g.hub_base);
                        word[k] = hrow[c >> 5];
                        sh[k] = c & 31;
                    }
                }
#pragma unroll
                for (int k = 0; k < CLQ_GATHERS; ++k) {
                    const int jb = j0 + k * GPM_WAVE;
                    if (jb >= Wall * 32) break;
                    const bool bit = live[k] && ((word[k] >> sh[k]) & 1u);
                    unsigned long long mask = __ballot(bit);
                    if (STATS && K >= 4 && bit) bytes3 += per_arc + 4ull * Dp[jb + lane];
                    if (K == 3) tri += bit;
                    if (K != 3 && lane == 0) {
                        stage[jb >> 5] = (uint32_t)mask;
                        stage[(jb >> 5) + 1] = (uint32_t)(mask >> 32);
                    }
                }
            }
        } else if (!ALL_HUB) {
            stage[lane] = 0;
            stage[GPM_WAVE + lane] = 0;
            wave_lds_sync();
            const int64_t vb = mid[v], ve = off[v + 1];
            for (int64_t p0 = vb; p0 < ve; p0 += 4 * GPM_WAVE) {  // four list reads in flight per lane
                int32_t w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t p = p0 + k * GPM_WAVE + lane;
                    w[k] = p < ve ? adj[p] : -1;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (w[k] < 0) continue;
                    int j = clq_hash_find(htab, hmask, hshift, U, w[k]);
                    if (j >= 0) {
                        atomicOr(stage + (j >> 5), 1u << (j & 31));
                        if (STATS && K >= 4) bytes3 += per_arc + 4ull * Dp[j];
                        if (K == 3) tri += 1;
                    }
                }
            }
        }
        if (K != 3) {
            wave_lds_sync();
            for (int t = 0; t < T; ++t) {
                const int Wt = big_words(d, t);
                if (i < big_rows(d, t) && lane < Wt) region[big_off(t) + (int64_t)i * Wt + lane] = stage[32 * t + lane];
            }
        }
        wave_lds_sync();
    }
    if (STATS) {  // merge-model bytes: level 2 of the rows built here, level 1 once per root
        for (int i = i0 + tid; i < i_end; i += 256) bytes += 8ull + 4ull * ((uint64_t)d + Dp[i]);
        if (i0 == 0 && tid == 0) bytes += 4ull + 4ull * (uint64_t)d;
        bytes = wave_sum_u64(bytes);
        bytes3 = wave_sum_u64(bytes3);
    }
    if (K == 3) tri = wave_sum_u64(tri);
    if (lane == 0) {
        if (STATS && bytes) atomicAdd(alg_bytes + (blockIdx.x & (CLQ_SLOTS - 1)), (unsigned long long)bytes);
        if (STATS && bytes3) atomicAdd(alg_bytes3 + (blockIdx.x & (CLQ_SLOTS - 1)), (unsigned long long)bytes3);
        if (K == 3 && tri) atomicAdd(n_matches + (blockIdx.x & (CLQ_SLOTS - 1)), (unsigned long long)tri);
    }
}

// The same rows built from the adjacency lists instead of the hub matrix:
// row i = N+(U[i]) ^ U by streaming the list N+(U[i]) with coalesced reads and
// testing every entry against a bitmap of U held in LDS (one LDS read and a
// shift per entry).  The position in U of an entry that passes is its rank
// among the set bits: a prefix count per bitmap word plus a popcount, so a hit
// costs one more LDS read, no search.  A gathered bit costs a 128-byte line of
// the hub matrix unless neighbouring lanes share it, which for universes of a
// few hundred vertices they rarely do; a list entry costs 4 bytes.
//
// The bitmap covers the ids [base_id, base_id + 32 * bm_words): the host passes
// the lowest root of the chunk as base_id (universes only hold vertices above
// their root) and sizes the LDS for it.
//
// grid = nroots * RB workgroups of NW waves; workgroup (ri, rb) builds the
// rows i = rb (mod RB) of root ri, its waves taking them in turn.  The bitmap
// and its prefix counts take up to 96 KiB, so one workgroup fits a compute unit
// whatever its size: NW = 16 gives the unit 16 waves to stream lists with.  The
// list ranges of 64 rows are fetched at once (one row per lane); a row's first
// 512 entries are in flight while the row before it is tested.
//
// Rows near the end of a universe belong to its highest-degree vertices: long
// lists, and few positions above the row left to test.  A row whose list holds
// at least `kappa4` / 4 entries per position above it is gathered from the hub
// matrix instead (every vertex of the launch's universes is a hub), one bit
// per position; kappa4 < 0: never.
// LDS (words): bitmap[bm_words] | prefix[bm_words / 4] (16-bit, one per word pair) | U[dcap] | G[dcap + 256] | stage[NW][128].
// x with lane LANE replaced by the scalar value v
template <int LANE>
__device__ __forceinline__ uint32_t write_lane(uint32_t x, uint32_t v) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(x) : "s"(v), "n"(LANE));
    return x;
}

// bit (v & 31) of word as one instruction: the bit-field extract takes its offset modulo 32
__device__ __forceinline__ uint32_t bit_of_word(uint32_t word, uint32_t v) {
    uint32_t h;
    asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(h) : "v"(word), "v"(v));
    return h;
}

#define CLQ_BM_MAX_WORDS 16384  // ids a bitmap can cover: 2^19
#define CLQ_LIST_LOADS 8        // list reads in flight per lane and row

template <int K, int NW>
__global__ __launch_bounds__(NW * GPM_WAVE) void k_clq_build_lists(GraphView g, const int32_t *roots, int64_t nroots,
                                                                 int dcap, int RB, int32_t base_id, int bm_words,
                                                                 int kappa4, int probe, uint32_t *rows_buf,
                                                                 int64_t stride_words, unsigned long long *n_matches) {
    // probe (a diagnostic, the counts are wrong with it), bits: 1 = the lists are read and nothing is tested,
    // 2 = the same with half of every list (what the delivery of the lists alone costs), 4 = rows that would be
    // gathered from the hub matrix are left empty, 8 = no row is written out
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t s_wave_sum[NW];
    constexpr int NT = NW * GPM_WAVE;
    // the wave number read as a scalar: row numbers, list ranges and loop tests then stay out of the vector unit
    const int tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid / GPM_WAVE);
    const int64_t ri = blockIdx.x / RB;
    const int rb = (int)(blockIdx.x % RB);
    if (ri >= nroots) return;
    const int64_t *off = g.off[0], *mid = g.mid[0];
    const int32_t *adj = g.adj[0];
    const int32_t u = roots[ri];
    const int64_t ub = mid[u];
    const int d = (int)(off[u + 1] - ub);
    if (rb >= d) return;
    uint32_t *bm = smem;                                                  // bm_words is a multiple of 1024
    uint16_t *prefix = reinterpret_cast<uint16_t *>(smem + bm_words);     // set bits before each pair of bitmap words
    int32_t *U = (int32_t *)(smem + bm_words + bm_words / 4);
    // G[j]: where the hub-matrix bit of column U[j] lies in a row: bit number (bits 0..4: a bit-field extract
    // takes its offset from there) and byte offset of its word (from bit 5 up), worked out once per root for
    // all the rows that are gathered.  G has dcap + 256 entries: the 256 after d stand for the root itself,
    // so a gather past the universe reads a valid place (its bit is dropped).
    uint32_t *G = smem + bm_words + bm_words / 4 + dcap;
    uint32_t *stage = smem + bm_words + bm_words / 4 + 2 * dcap + CLQ_GATHERS * GPM_WAVE + wave * 128;
    for (int k = tid; k < bm_words / 4; k += NT) reinterpret_cast<uint4 *>(bm)[k] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (int i = tid; i < d; i += NT) {
        const int32_t v = adj[ub + i];
        U[i] = v;
        const uint32_t c = (uint32_t)(v - base_id);
        atomicOr(bm + (c >> 5), 1u << (c & 31));
    }
    for (int i = tid; i < d + CLQ_GATHERS * GPM_WAVE; i += NT) {  // a trip of gathers begins below d and reads 256 entries
        const uint32_t h = (uint32_t)((i < d ? adj[ub + i] : u) - g.hub_base);
        G[i] = ((h >> 5) << 7) | (h & 31u);
    }
    __syncthreads();
    {  // prefix counts: every thread sums its run of words, the runs are scanned, the words filled in
        const int wpt = bm_words / NT;  // bm_words is a multiple of 1024
        uint32_t mine = 0;
        for (int k = 0; k < wpt; ++k) mine += __popc(bm[tid * wpt + k]);
        uint32_t incl = mine;
        for (int o = 1; o < GPM_WAVE; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, GPM_WAVE);
            if (lane >= o) incl += up;
        }
        if (lane == GPM_WAVE - 1) s_wave_sum[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (int w = 0; w < wave; ++w) before += s_wave_sum[w];
        for (int k = 0; k < wpt; ++k) {
            const int w = tid * wpt + k;
            if (!(w & 1)) prefix[w >> 1] = (uint16_t)before;
            before += __popc(bm[w]);
        }
    }
    __syncthreads();

    const int T = big_tiles(d);
    uint32_t *region = rows_buf + ri * stride_words;
    uint64_t tri = 0;       // per lane
    uint32_t tri_wave = 0;  // the same for every lane: bits counted from ballots
    const int step = NW * RB;              // distance between two rows of one wave
    const int first = rb + RB * wave;      // this wave's first row
    for (int i_batch = first; i_batch < d; i_batch += 64 * step) {
        // list ranges of the next 64 rows of this wave, one row per lane; n = -1: the row is gathered
        int64_t my_b = 0;
        int my_n = 0;
        {
            const int i_l = i_batch + lane * step;
            if (i_l < d) {
                const int32_t v = U[i_l];
                my_b = mid[v];
                my_n = (int)(off[v + 1] - my_b);
                if (probe & 2) my_n = (my_n + 1) / 2;
                if (kappa4 >= 0 && (int64_t)(d - 1 - i_l) * kappa4 <= 4 * (int64_t)my_n) my_n = -1;
            }
        }
        const uint32_t my_b_lo = (uint32_t)my_b, my_b_hi = (uint32_t)((uint64_t)my_b >> 32);
        auto range_of = [&](int r, int64_t &b, int &n) {
            b = (int64_t)(((uint64_t)__builtin_amdgcn_readlane(my_b_hi, r) << 32) | __builtin_amdgcn_readlane(my_b_lo, r));
            n = __builtin_amdgcn_readlane(my_n, r);
        };
        int32_t cur[CLQ_LIST_LOADS], nxt[CLQ_LIST_LOADS];
        int64_t b;
        int n;
        range_of(0, b, n);
        // a lane past the end of a list holds the root itself: it lies inside the bitmap's range and no
        // universe holds its own root, so it fails the test like any other foreign vertex
#pragma unroll
        for (int k = 0; k < CLQ_LIST_LOADS; ++k) {
            const int p = k * GPM_WAVE + lane;
            cur[k] = p < n ? adj[b + p] : u;
            nxt[k] = u;
        }
        for (int r = 0; r < 64; ++r) {
            const int i = i_batch + r * step;
            if (i >= d) break;
            int64_t nb = 0;
            int nn = 0;
            if (r + 1 < 64 && i + step < d) range_of(r + 1, nb, nn);
            if (nn > 0) {  // (a gathered or empty next row has nothing to ask for: one scalar branch instead of eight masked reads)
#pragma unroll
                for (int k = 0; k < CLQ_LIST_LOADS; ++k) {  // the next row's entries are in flight while this row is tested
                    const int p = k * GPM_WAVE + lane;
                    nxt[k] = p < nn ? adj[nb + p] : u;
                }
            }
            if (K != 3) {
                stage[lane] = 0;
                stage[GPM_WAVE + lane] = 0;
                wave_lds_sync();
            }
            // the bitmap words of a batch of entries are all requested before the first is looked at
            auto hit = [&](uint32_t c, uint32_t word) {
                if (K == 3) {
                    tri += 1;
                } else {
                    // rank among the set bits = position in U: the count before the word pair, plus the pair's bits below
                    const uint64_t pair = reinterpret_cast<const uint64_t *>(bm)[c >> 6];
                    const int j = (int)prefix[c >> 6] + __popcll(pair & ((1ull << (c & 63)) - 1ull));
                    atomicOr(stage + (j >> 5), 1u << (j & 31));
                }
            };
            if (n < 0 && !(probe & 4)) {  // a row of few positions and a long list: one hub-matrix bit per position above it
                // The row's address is a scalar and a lane's place in it comes ready from G: a gather is one LDS
                // read, one shift, one read of global memory, one bit-field extract and a compare.  Which
                // positions count (above i, below d) is tested per lane, and only in a row's first and last
                // trip; the ballots are scalars as they come, and the four masks of a trip go to the stage
                // in one write.
                const int32_t vi = __builtin_amdgcn_readfirstlane(U[i]);
                const char *hrow = reinterpret_cast<const char *>(g.hub + (size_t)(vi - g.hub_base) * (size_t)g.hub_wpr);
                const int jfirst = (i + 1) & ~(GPM_WAVE - 1);
                for (int j0 = jfirst; j0 < d; j0 += CLQ_GATHERS * GPM_WAVE) {
                    uint32_t gx[CLQ_GATHERS], word[CLQ_GATHERS], bit[CLQ_GATHERS];
#pragma unroll
                    for (int k = 0; k < CLQ_GATHERS; ++k) gx[k] = G[j0 + k * GPM_WAVE + lane];
                    // the lanes of the first 64 that lie at or below i read the row's first word instead of a scattered one
                    if (j0 == jfirst && j0 + lane <= i) gx[0] = 0;
#pragma unroll
                    for (int k = 0; k < CLQ_GATHERS; ++k)
                        word[k] = *reinterpret_cast<const uint32_t *>(hrow + (gx[k] >> 5));
#pragma unroll
                    for (int k = 0; k < CLQ_GATHERS; ++k) bit[k] = bit_of_word(word[k], gx[k]);
                    if (j0 == jfirst || j0 + CLQ_GATHERS * GPM_WAVE > d) {
#pragma unroll
                        for (int k = 0; k < CLQ_GATHERS; ++k) {
                            const int j = j0 + k * GPM_WAVE + lane;
                            if (j <= i || j >= d) bit[k] = 0;
                        }
                    }
                    uint32_t out = 0;  // lane l: word l of the eight the trip's masks fill
                    const unsigned long long m0 = __ballot(bit[0] != 0u), m1 = __ballot(bit[1] != 0u),
                                             m2 = __ballot(bit[2] != 0u), m3 = __ballot(bit[3] != 0u);
                    static_assert(CLQ_GATHERS == 4, "four masks per trip");
                    if (K == 3) tri_wave += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
                    out = write_lane<0>(out, (uint32_t)m0);
                    out = write_lane<1>(out, (uint32_t)(m0 >> 32));
                    out = write_lane<2>(out, (uint32_t)m1);
                    out = write_lane<3>(out, (uint32_t)(m1 >> 32));
                    out = write_lane<4>(out, (uint32_t)m2);
                    out = write_lane<5>(out, (uint32_t)(m2 >> 32));
                    out = write_lane<6>(out, (uint32_t)m3);
                    out = write_lane<7>(out, (uint32_t)(m3 >> 32));
                    // the stage has room for 4096 positions; words past it would be zeros anyway
                    if (K != 3 && lane < 2 * CLQ_GATHERS && (j0 >> 5) + lane < 128) stage[(j0 >> 5) + lane] = out;
                }
            }
            if (n > 0 && (probe & 3)) {
                uint32_t x = 0;
#pragma unroll
                for (int k = 0; k < CLQ_LIST_LOADS; ++k) x ^= (uint32_t)cur[k];
                tri += x;
            } else if (n > 0) {
                uint32_t c[CLQ_LIST_LOADS], word[CLQ_LIST_LOADS];
#pragma unroll
                for (int k = 0; k < CLQ_LIST_LOADS; ++k) {
                    c[k] = (uint32_t)(cur[k] - base_id);
                    word[k] = bm[c[k] >> 5];
                }
#pragma unroll
                for (int k = 0; k < CLQ_LIST_LOADS; ++k)
                    if ((word[k] >> (c[k] & 31)) & 1u) hit(c[k], word[k]);
            }
            for (int p0 = CLQ_LIST_LOADS * GPM_WAVE; p0 < n; p0 += 4 * GPM_WAVE) {  // the rest of a long list
                uint32_t c[4], word[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int p = p0 + k * GPM_WAVE + lane;
                    c[k] = (uint32_t)((p < n ? adj[b + p] : u) - base_id);
                }
                if (probe & 3) {
                    tri += c[0] ^ c[1] ^ c[2] ^ c[3];
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) word[k] = bm[c[k] >> 5];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((word[k] >> (c[k] & 31)) & 1u) hit(c[k], word[k]);
            }
            if (K != 3 && !(probe & 8)) {
                wave_lds_sync();
                for (int t = 0; t < T; ++t) {
                    const int Wt = big_words(d, t);
                    if (i < big_rows(d, t) && lane < Wt) region[big_off(t) + (int64_t)i * Wt + lane] = stage[32 * t + lane];
                }
                wave_lds_sync();
            }
#pragma unroll
            for (int k = 0; k < CLQ_LIST_LOADS; ++k) cur[k] = nxt[k];
            b = nb;
            n = nn;
        }
    }
    if (K != 3 && probe && tri == 0x12345678abcdULL) region[0] = 1;  // keeps the probe's loads alive
    if (K == 3) {
        tri = wave_sum_u64(tri);
        if (lane == 0 && tri + tri_wave) atomicAdd(n_matches + (blockIdx.x & (CLQ_SLOTS - 1)), (unsigned long long)(tri + tri_wave));
    }
}

// Inclusive prefix sum over the 64 lanes of a wave with data-parallel-primitive adds: four shifts inside
// the rows of 16 lanes, then the row totals carried into the rows above.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return x;
}

// grid = nroots * phases * SEG workgroups of NW waves (phases = 1 for
// dcap <= 1024, else the tile pairs).  LDS (words): J[NW][512] (16-bit entries)
// | tile rows.
//
// For every row i the wave first lists the set bits j of row i over the row
// tile (the rows j to AND with), then lane blocks of L lanes each take entries
// of the list, read row j's words over the column tile from LDS, 16 bytes per
// lane, and add |row_i ^ row_j|.
//
// Listing.  Lane l holds 16 of the row's 1024 positions (four bits of four words spread over the row); an inclusive scan of
// the per-lane bit counts over the wave gives every lane the place of its first
// entry, and each lane then writes its own entries (the order inside a list
// does not matter).  List entries are row offsets in 16-byte slots (j * L), so
// a row's address is one shift-add.
//
// LDS layout and lane mapping.  A tile row is held at a stride of L slots,
// L = lanes per row = its words / 4 rounded up to a power of two.  The LDS
// serves a 16-byte-per-lane read in four groups of 16 lanes, each group one
// cycle when its lanes touch 16 different slots (slot = 16-byte address mod
// 16).  With L = 8 (the 1024-column tiles) a group holds half rows of four lane
// blocks, two lower and two upper halves, and two rows clash exactly when they
// have the same parity: the set bits are therefore listed by parity, and of the
// four blocks of a group two always read even rows and two odd rows.  Each
// parity list is read by four blocks per instruction.  Narrower tiles (L < 8)
// use one list.
template <int SEG, int NW>
__global__ __launch_bounds__(NW * GPM_WAVE) void k_clq_count4(GraphView g, const int32_t *roots, int64_t nroots, int dcap,
                                                    int vt_words, int probe, const uint32_t *rows_buf,
                                                    int64_t stride_words, unsigned long long *n_matches) {
    // probe (a diagnostic, the counts are wrong with it), bits: 1 = tiles are copied in and set bits listed, no
    // rows are ANDed (what everything but the ANDs costs); 2 = the set bits are counted and scanned but no
    // list entry is written (with 1: what the emission loops cost)
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // the wave number read as a scalar: row numbers, row addresses and loop tests then stay out of the vector unit
    const int tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid / GPM_WAVE);
    // The count runs over tiles of VT = 32 * vt_words columns (256, 512 or 1024) cut out of the stored tiles
    // of 1024: row j's bits lie above j, so against a row j of tile a only the tiles b >= a hold
    // anything, and the narrower the tiles, the fewer empty words of the rows j near a universe's end
    // (where the bits of a row crowd) are read and ANDed.  Narrow tiles also take less LDS: more
    // workgroups share a compute unit, and one's copy into LDS runs under the others' counting.
    const int VT = 32 * vt_words;
    const int tcap = dcap > VT ? dcap / VT : 1;      // tiles a root of this class can have
    const int phases = tcap * (tcap + 1) / 2;        // tile pairs (a, b), a <= b
    const int64_t ri = blockIdx.x / (phases * SEG);
    const int ph = (int)(blockIdx.x / SEG) % phases;
    const int seg = (int)(blockIdx.x % SEG);
    if (ri >= nroots) return;
    const int32_t u = roots[ri];
    const int d = (int)(g.off[0][u + 1] - g.mid[0][u]);
    int ta = 0, tb = ph;  // the ph-th pair in the order (0,0) (0,1) .. (0,tcap-1) (1,1) ..
    while (tb >= tcap - ta) {
        tb -= tcap - ta;
        ta++;
    }
    tb += ta;
    if (VT * tb >= d) return;
    const uint32_t *region = rows_buf + ri * stride_words;

    // what this task holds in LDS and walks: the rows j of tile a restricted to the columns of tile b;
    // rows i < j_hi are walked, their bits over tile a list the j, their words over tile b are ANDed
    const int j_lo = VT * ta, j_hi = min(d, VT * (ta + 1)), i_hi = j_hi;
    // stored tile and word offset inside its rows of the two tiles
    const int sa = (VT * ta) >> 10, wa = ((VT * ta) & 1023) >> 5, sb = (VT * tb) >> 10, wb = ((VT * tb) & 1023) >> 5;
    const int Wsa = big_words(d, sa), Wsb = big_words(d, sb);  // words of a stored row
    const int Wt = min(vt_words, Wsb - wb), Wj = min(vt_words, Wsa - wa);
    const uint32_t *xsrc = region + big_off(sb) + wb;            // row i's words over the column tile, rows Wsb apart
    const uint32_t *jsrc = region + big_off(sa) + wa;            // row i's words over the row tile's positions, rows Wsa apart
    const uint32_t *tile_rows = xsrc + (size_t)j_lo * Wsb;       // rows j_lo.. of the column tile
    uint16_t *J = (uint16_t *)smem + wave * VT;                  // a list holds at most VT entries
    uint32_t *tile = smem + NW * (VT / 2);
    const int lg = Wt <= 4 ? 0 : Wt <= 8 ? 1 : Wt <= 16 ? 2 : 3;  // Wt <= 32
    const int L = 1 << lg, sub = lane & (L - 1), blk = lane >> lg;
    const bool lane_ok = 4 * sub < Wt;
    {
        // Rows copied in at a stride of L slots; the slots past a row's Wt / 4 are never read.  The copy runs
        // over the nrows * Wq 16-byte words to fetch: every read of a full batch is issued without a
        // condition, so eight of them are in flight per thread.  Only one workgroup fits a compute unit
        // with the big tiles, and nothing else hides this copy's latency.
        const int Wq = Wt >> 2, Wsq = Wsb >> 2, nsrc = (j_hi - j_lo) * Wq;
        const uint32_t inv = (65536u + (uint32_t)Wq - 1u) / (uint32_t)Wq;  // q / Wq == (q * inv) >> 16 for q < 8192, Wq <= 8
        const uint4 *src = reinterpret_cast<const uint4 *>(tile_rows);
        uint4 *dst = reinterpret_cast<uint4 *>(tile);
        constexpr int NT = NW * GPM_WAVE, FILL = 8;
        int row, col;
        auto split = [&](int q) {
            row = (int)(((uint32_t)q * inv) >> 16);
            col = q - row * Wq;
        };
        int q0 = tid;
        for (; q0 + (FILL - 1) * NT < nsrc; q0 += FILL * NT) {
            uint4 v[FILL];
#pragma unroll
            for (int f = 0; f < FILL; ++f) {
                split(q0 + f * NT);
                v[f] = src[row * Wsq + col];
            }
#pragma unroll
            for (int f = 0; f < FILL; ++f) {
                split(q0 + f * NT);
                dst[(row << lg) + col] = v[f];
            }
        }
        for (; q0 < nsrc; q0 += NT) {
            split(q0);
            dst[(row << lg) + col] = src[row * Wsq + col];
        }
    }
    __syncthreads();

    const bool two = lg == 3;  // two lists, by parity of j
    // reading: list and first entry of this lane block, entries per step
    const int my_res = two ? (blk >> 1) & 1 : 0;
    const int my_first = two ? (blk & 1) + 2 * (blk >> 2) : blk;
    const int my_step = two ? 4 : GPM_WAVE >> lg;
    // The even list grows up from J[0], the odd list down from J[1023] (together they hold at most 1023
    // entries): entry t of one and entry t of the other then never lie on the same LDS bank, which they
    // would if the second list started half way (1024 bytes on).
    // this lane block's first entry and the distance (in entries, signed) between two of its entries
    const uint16_t *Jfirst = my_res ? J + VT - 1 - my_first : J + my_first;
    const int jstep = my_res ? -my_step : my_step;
    // listing: lane l looks at four bits of each of the words (l & 7) + 8 k, k = 0..3: bits 4 (l >> 3) .. + 3.
    // A row is dense where its universe holds the highest-degree vertices (its last positions), and a lane
    // that held 16 neighbouring positions of that region would write its entries long after the others were
    // done; this way every lane gets a share of every quarter of the row.
    const int word_of_lane = lane & 31;
    const int wg = lane & 7, nib = lane >> 3;
    const uint32_t pos0 = (uint32_t)(32 * wg + 4 * nib);
    const uint32_t *lane_base = tile + 4 * sub;

    uint64_t total = 0;
    const int step = SEG * NW;
    int i = seg * NW + wave;
    // row i's list words (lanes l and l + 32 hold word l) and its column-tile words, fetched one row ahead
    uint32_t mine = 0;
    u32x4 C = {0u, 0u, 0u, 0u};
    if (i < i_hi) {
        if (word_of_lane < Wj) mine = jsrc[(size_t)i * Wsa + word_of_lane];
        if (lane_ok) C = load4(xsrc + (size_t)i * Wsb + 4 * sub);
    }
    while (i < i_hi) {
        // list the set bits of row i over tile a
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            bits |= (((uint32_t)__shfl((int)mine, wg + 8 * k, GPM_WAVE) >> (4 * nib)) & 0xfu) << (4 * k);
        const uint32_t c = two ? (uint32_t)__popc(bits & 0x5555u) | ((uint32_t)__popc(bits & 0xaaaau) << 16)
                               : (uint32_t)__popc(bits);
        const uint32_t incl = wave_incl_scan_u32(c);
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, GPM_WAVE - 1);
        // next free entry, even / odd list.  Bit b of `bits` is position pos0 + 256 (b >> 2) + (b & 3) (word
        // (b >> 2) * 8 further on), an entry is that position's slot offset.  One loop per list keeps the
        // parity out of the loop body.
        uint16_t *at0 = J + ((incl - c) & 0xffffu), *at1 = J + VT - 1 - ((incl - c) >> 16);
        const uint32_t entry0 = pos0 << lg;
        uint32_t be = two ? bits & 0x5555u : bits, bo = two ? bits & 0xaaaau : 0u;
        if (probe & 2) be = bo = 0;  // no entries written
        while (be) {
            const uint32_t b = (uint32_t)__builtin_ctz(be);
            be &= be - 1;
            *at0++ = (uint16_t)(entry0 + ((((b & 12u) << 6) | (b & 3u)) << lg));
        }
        while (bo) {
            const uint32_t b = (uint32_t)__builtin_ctz(bo);
            bo &= bo - 1;
            *at1-- = (uint16_t)(entry0 + ((((b & 12u) << 6) | (b & 3u)) << lg));
        }
        // the next row's words are requested once this row's have been used
        const int inext = i + step;
        uint32_t mine_n = 0;
        u32x4 C_n = {0u, 0u, 0u, 0u};
        if (inext < i_hi) {
            if (word_of_lane < Wj) mine_n = jsrc[(size_t)inext * Wsa + word_of_lane];
            if (lane_ok) C_n = load4(xsrc + (size_t)inext * Wsb + 4 * sub);
        }
        const int my_len = (int)(my_res ? all >> 16 : all & 0xffffu);
        wave_lds_sync();
        if (all && lane_ok && !probe) {
            uint32_t acc = 0;
            int t = my_first;
            const uint16_t *jp = Jfirst;  // entry t of this block's list
            for (; t + 3 * my_step < my_len; t += 4 * my_step, jp += 4 * jstep) {  // four independent row reads in flight
                const u32x4 r0 = load4(lane_base + ((uint32_t)jp[0] << 2));
                const u32x4 r1 = load4(lane_base + ((uint32_t)jp[jstep] << 2));
                const u32x4 r2 = load4(lane_base + ((uint32_t)jp[2 * jstep] << 2));
                const u32x4 r3 = load4(lane_base + ((uint32_t)jp[3 * jstep] << 2));
                acc = and_popc_add4(C, r0, acc);
                acc = and_popc_add4(C, r1, acc);
                acc = and_popc_add4(C, r2, acc);
                acc = and_popc_add4(C, r3, acc);
            }
            for (; t < my_len; t += my_step, jp += jstep) acc = and_popc_add4(C, load4(lane_base + ((uint32_t)jp[0] << 2)), acc);
            total += acc;
        }
        wave_lds_sync();
        mine = mine_n;
        C = C_n;
        i = inext;
    }
    if (probe) total += J[lane];  // keeps the probe's lists alive
    total = wave_sum_u64(total);
    if (lane == 0 && total) atomicAdd(n_matches + (blockIdx.x & (CLQ_SLOTS - 1)), (unsigned long long)total);
}

// ---------------------------------------------------------------------------
// 4-cycles: two-path counting
// ---------------------------------------------------------------------------
//
// On a simple symmetric graph every 4-cycle has one largest vertex v1 (ids are
// degree ranks, so v1 is its highest-degree vertex) and one vertex v3 opposite
// to it; the other two, v2 and v4, are common neighbours of v1 and v3 that lie
// below v1.  With c(v1, v3) = number of two-paths v1 - v2 - v3 whose v2 and v3
// are both below v1, the cycles with largest vertex v1 number
// sum over v3 of c(c-1)/2, and each stands for 8 embeddings (the reference
// reports every automorphic image, reference: src/gpm-par.c:179-243).  Walking
// only below v1 keeps the two-path count near sum over arcs of the smaller
// degree, whatever the skew.
//
// The counters c(v1, .) live in a dense array of n 32-bit words that the
// walking workgroups own; atomicAdd returns the old value k and adding up all
// old values gives sum c(c-1)/2 without a second reading pass.  The array is
// zero between roots: light roots walk their two-paths again and store zeros,
// heavy roots (all workgroups on one root) are followed by a memset of [0, v1).
//
// Work is balanced whatever the skew: wp[p] is the number of two-paths of all
// arcs before arc p (prefix sum over the CSR, arcs to a higher vertex count
// zero), so the two-paths of root v1 are the range [wp[off[v1]], wp[mid[v1]])
// and a wave takes a block of CYC_BLOCK consecutive two-paths of it, finding
// the arc it starts in by binary search and then walking tiles of 64 arcs.

#define CYC_BLOCK 8192

// Two-paths of every arc p = (v1 -> v2): the entries of N(v2) below v1 when
// v2 < v1, else none.  One thread per arc.
__global__ void k_cycle4_arc_paths(GraphView g, int64_t m, int64_t *len) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int64_t *off = g.off[0];
    const int32_t *adj = g.adj[0];
    int32_t lo = 0, hi = g.n;  // last v with off[v] <= p
    while (hi - lo > 1) {
        int32_t mid = lo + ((hi - lo) >> 1);
        if (off[mid] <= p) lo = mid;
        else hi = mid;
    }
    const int32_t v1 = lo, v2 = adj[p];
    len[p] = v2 < v1 ? lower_bound_g(adj, off[v2], off[v2 + 1], v1) - off[v2] : 0;
}

// cost[i] = two-paths of root index i (vertex n-1-i), plus one.
__global__ void k_cycle4_cost(GraphView g, const int64_t *wp, int64_t nroots, int64_t *cost) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nroots) return;
    const int32_t v1 = (int32_t)(g.n - 1 - i);
    cost[i] = wp[g.mid[0][v1]] - wp[g.off[0][v1]] + 1;
}

// arcs[i] = arcs of root index idx[i] (vertex n-1-idx[i]) that lead below it.
__global__ void k_cycle4_root_arcs(GraphView g, const int64_t *idx, int64_t count, int64_t *arcs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int32_t v1 = (int32_t)(g.n - 1 - idx[i]);
    arcs[i] = g.mid[0][v1] - g.off[0][v1];
}

// One wave's pass over blocks blk0, blk0 + bstep, ... of the two-paths of root
// v1.  RESET = false: count, returns the sum of the old counter values;
// RESET = true: store zeros.  s_pre / s_beg are this wave's 64-entry LDS arrays.
//
// Hot counters.  A two-path of a heavy root lands, on average, on a counter
// that more than a hundred others land on as well (hub to hub pairs), and
// atomics on one address are served one after the other.  A workgroup that
// walks a long span of a root's two-paths therefore keeps a cache of counters
// in LDS: a table of CYC_CACHE (vertex, count) slots, two neighbouring slots per vertex; the first
// vertex to reach an empty slot owns it for the rest of the span, vertices
// that find their slot taken go to the global counter as before.  At the end
// of the span a slot with count k is added to the global counter in one
// atomic whose old value o stands for o*k + k(k-1)/2 new pairs of two-paths.
#define CYC_CACHE 16384
#define CYC_CACHE_BITS 14

__device__ __forceinline__ uint32_t cyc_count_cached(uint32_t *cnt, int32_t *keys, uint32_t *cnts, int32_t v3) {
    uint32_t h = ((uint32_t)v3 * 2654435761u) >> (32 - CYC_CACHE_BITS);
#pragma unroll
    for (int way = 0; way < 2; ++way, h ^= 1u) {  // two neighbouring slots per vertex
        int32_t k = keys[h];
        if (k == -1) {
            k = atomicCAS(keys + h, -1, v3);
            if (k == -1) k = v3;
        }
        if (k == v3) {
            atomicAdd(cnts + h, 1u);
            return 0;
        }
    }
    return atomicAdd(cnt + v3, 1u);
}

template <bool RESET, bool CACHE = false>
__device__ __forceinline__ uint64_t cyc_wave_pass(const GraphView &g, const int64_t *wp, int32_t v1, int64_t blk0,
                                                  int64_t bstep, uint32_t *cnt, uint32_t *s_pre, int64_t *s_beg,
                                                  int64_t block = CYC_BLOCK, int64_t blk_end = INT64_MAX,
                                                  int32_t *keys = nullptr, uint32_t *cnts = nullptr) {
    const int lane = lane_id();
    const int64_t *off = g.off[0];
    const int32_t *adj = g.adj[0];
    const int64_t lb = off[v1], le = g.mid[0][v1];
    const int64_t base = wp[lb], W = wp[le] - base;
    const int64_t nblk = min(blk_end, (W + block - 1) / block);
    uint64_t acc = 0;
    for (int64_t blk = blk0; blk < nblk; blk += bstep) {
        int64_t w0 = blk * block;
        const int64_t w1 = min(W, w0 + block);
        int64_t a = lb, ae = le;  // last arc a in [lb, le) with wp[a] - base <= w0
        while (ae - a > 1) {
            int64_t mid = a + ((ae - a) >> 1);
            if (wp[mid] - base <= w0) a = mid;
            else ae = mid;
        }
        while (w0 < w1) {
            // tile of the 64 arcs from a on: their two-paths are [t0, t1)
            const int64_t q = a + lane;
            const int64_t t0 = wp[a] - base;
            const int64_t qe = min(le, a + GPM_WAVE);
            const int64_t t1 = wp[qe] - base;
            uint32_t pre = 0xffffffffu;
            int64_t beg = 0;
            if (q < qe) {
                pre = (uint32_t)(wp[q] - base - t0);
                beg = off[adj[q]];
            }
            wave_lds_sync();
            s_pre[lane] = pre;
            s_beg[lane] = beg;
            wave_lds_sync();
            const int64_t wend = min(w1, t1);
            for (int64_t wq = w0; wq < wend; wq += 4 * GPM_WAVE) {
                int32_t v3[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t w = wq + u * GPM_WAVE + lane;
                    v3[u] = -1;
                    if (w < wend) {
                        const uint32_t r = (uint32_t)(w - t0);
                        int j = 0;  // last entry with s_pre[j] <= r
#pragma unroll
                        for (int st = GPM_WAVE / 2; st > 0; st >>= 1)
                            if (s_pre[j + st] <= r) j += st;
                        v3[u] = adj[s_beg[j] + (r - s_pre[j])];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (v3[u] < 0) continue;
                    if (RESET) cnt[v3[u]] = 0;
                    else if (CACHE) acc += cyc_count_cached(cnt, keys, cnts, v3[u]);
                    else acc += atomicAdd(cnt + v3[u], 1u);
                }
            }
            w0 = wend;
            a = qe;
        }
    }
    return acc;
}

// Which kernel counts a root of `paths` two-paths.  Counters in LDS (k_cycle4_ranges) cost a fixed amount per
// range of S ids below the root, whatever the root's work, so a root takes them when it has at least
// `per_range` two-paths for every range (and `ranged_from` in all); other roots are cut into spans
// (k_cycle4_heavy) from `heavy_from` two-paths on, and below that one workgroup walks them (k_cycle4_light).
struct CycRule {
    int64_t heavy_from, ranged_from;
    int32_t S, per_range;  // S = 0: no root counts in LDS
};
__host__ __device__ __forceinline__ bool cyc_is_ranged(const CycRule &rule, int64_t paths, int32_t v1) {
    if (!rule.S || paths < rule.ranged_from) return false;
    return paths >= (int64_t)rule.per_range * (((int64_t)v1 + rule.S - 1) / rule.S);
}
__host__ __device__ __forceinline__ bool cyc_is_heavy(const CycRule &rule, int64_t paths, int32_t v1) {
    return paths >= rule.heavy_from || cyc_is_ranged(rule, paths, v1);
}

// Light roots: one workgroup of 4 waves per root, roots taken from a queue
// (index order = heaviest first); workgroup b owns the counter array
// dense + b * stride.  Roots that `rule` calls heavy
// are left to k_cycle4_heavy.
#define CYC_HEAVY_GROUPS 3  // groups of counter arrays for heavy roots, one group per stream
__global__ __launch_bounds__(256) void k_cycle4_light(GraphView g, const int64_t *wp, int64_t lo, int64_t hi,
                                                      const int64_t *cost, CycRule rule,
                                                      unsigned long long *queue, uint32_t *dense, int64_t stride,
                                                      unsigned long long *n_cycles, unsigned long long *alg_bytes) {
    __shared__ uint32_t s_pre[4][GPM_WAVE];
    __shared__ int64_t s_beg[4][GPM_WAVE];
    __shared__ long long s_next;
    const int wave = threadIdx.x / GPM_WAVE, lane = lane_id();
    uint32_t *cnt = dense + (size_t)blockIdx.x * (size_t)stride;
    uint64_t acc = 0, bytes = 0;
    // roots are claimed in batches: one atomic per root on a single word would serialise a chunk of a
    // million light roots
    const int64_t batch = max((int64_t)1, min((int64_t)256, (hi - lo) / ((int64_t)gridDim.x * 8)));
    int64_t i = 0, i_end = 0;
    for (;;) {
        if (i == i_end) {
            __syncthreads();
            if (threadIdx.x == 0) s_next = (long long)(lo + (int64_t)atomicAdd(queue, (unsigned long long)batch));
            __syncthreads();
            i = s_next;
            if (i >= hi) break;
            i_end = min(hi, i + batch);
        }
        if (i == s_next && batch > 1) {
            // first the tiny roots of the batch (at most 64 two-paths), one wave each and no counter
            // array: every lane takes one two-path and counts the lower lanes that end in the same vertex
            for (int64_t t = i + wave; t < i_end; t += 4) {
                const int64_t np = cost[t] - 1;
                if (np < 2 || np > GPM_WAVE) continue;
                const int32_t r1 = (int32_t)(g.n - 1 - t);
                const int64_t lb = g.off[0][r1], le = g.mid[0][r1];
                const int64_t base = wp[lb];
                int32_t v3 = -1;
                if (lane < np) {
                    int64_t a = lb, ae = le;  // last arc with wp[a] - base <= lane
                    while (ae - a > 1) {
                        int64_t mid = a + ((ae - a) >> 1);
                        if (wp[mid] - base <= lane) a = mid;
                        else ae = mid;
                    }
                    v3 = g.adj[0][g.off[0][g.adj[0][a]] + (lane - (wp[a] - base))];
                }
                uint32_t same = 0;
                for (int l = 0; l < (int)np - 1; ++l) {
                    const int32_t x = __builtin_amdgcn_readlane(v3, l);
                    same += l < lane && x == v3;
                }
                acc += same;
                if (lane == 0) bytes += 8ull * (uint64_t)np + 4ull + 12ull * (uint64_t)(le - lb);
            }
        }
        const int64_t paths = cost[i] - 1;
        if (paths < 2 || (batch > 1 && paths <= GPM_WAVE) || cyc_is_heavy(rule, paths, (int32_t)(g.n - 1 - i))) {  // a cycle needs two two-paths
            ++i;
            continue;
        }
        // The zeros of the last root are in place before this root counts.  The array belongs to this
        // workgroup alone, so workgroup scope is enough (the barrier carries it): stores and atomics both
        // complete in L2, and a wider fence would write the whole L2 back once per root.
        __syncthreads();
        const int32_t v1 = (int32_t)(g.n - 1 - i);
        // four blocks at least, so that every wave of the workgroup has two-paths to walk
        const int64_t blk = min((int64_t)CYC_BLOCK, max((int64_t)256, (((paths + 3) >> 2) + 255) & ~(int64_t)255));
        acc += cyc_wave_pass<false>(g, wp, v1, wave, 4, cnt, s_pre[wave], s_beg[wave], blk);
        // bytes of this root: its list and one prefix per arc, every two-path read and counted
        if (threadIdx.x == 0) bytes += 8ull * (uint64_t)paths + 4ull + 12ull * (uint64_t)(g.mid[0][v1] - g.off[0][v1]);
        __syncthreads();
        if (paths * 8 >= (int64_t)v1) {  // zero [0, v1) with coalesced stores
            uint4 *z = reinterpret_cast<uint4 *>(cnt);
            for (int x = threadIdx.x; x < (v1 + 3) / 4; x += 256) z[x] = make_uint4(0u, 0u, 0u, 0u);
        } else {
            cyc_wave_pass<true>(g, wp, v1, wave, 4, cnt, s_pre[wave], s_beg[wave], blk);
        }
        ++i;
    }
    acc = wave_sum_u64(acc);
    bytes = wave_sum_u64(bytes);
    if (lane == 0) {
        if (acc) atomicAdd(n_cycles, (unsigned long long)acc);
        if (bytes) atomicAdd(alg_bytes, (unsigned long long)bytes);
    }
}

// Heavy roots, up to CYC_BATCH of them per launch (fewer when memory for the arrays is short), each with a counter array of
// its own (root r of the batch: dense + r * stride; the caller zeroes them
// afterwards).  A root is cut into spans of `span` two-paths, one workgroup of
// 16 waves per span (workgroups wg0[r] .. wg0[r + 1] belong to root r); the
// waves share the span's blocks of `block` two-paths and one LDS counter cache.
// Dynamic LDS: s_beg[16][64] (int64) | s_pre[16][64] | keys[CYC_CACHE] | cnts[CYC_CACHE].
#define CYC_BATCH 128
#define CYC_HEAVY_WAVES 16
struct CycBatch {
    int32_t n;
    int32_t v1[CYC_BATCH];
    int32_t wg0[CYC_BATCH + 1];
};
#define CYC_HEAVY_LDS_BYTES (CYC_HEAVY_WAVES * GPM_WAVE * 12 + CYC_CACHE * 8)

__global__ __launch_bounds__(CYC_HEAVY_WAVES *GPM_WAVE) void k_cycle4_heavy(GraphView g, const int64_t *wp, CycBatch b,
                                                                           uint32_t *dense, int64_t stride,
                                                                           int64_t span, int64_t block,
                                                                           unsigned long long *n_cycles,
                                                                           unsigned long long *alg_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    int64_t *s_beg = reinterpret_cast<int64_t *>(smem);
    uint32_t *s_pre = smem + CYC_HEAVY_WAVES * GPM_WAVE * 2;
    int32_t *keys = reinterpret_cast<int32_t *>(s_pre + CYC_HEAVY_WAVES * GPM_WAVE);
    uint32_t *cnts = reinterpret_cast<uint32_t *>(keys) + CYC_CACHE;
    const int wave = threadIdx.x / GPM_WAVE, lane = lane_id();
    int r = 0;  // last root with wg0[r] <= blockIdx.x
    for (int hi = b.n; hi - r > 1;) {
        const int m = (r + hi) >> 1;
        if (b.wg0[m] <= (int)blockIdx.x) r = m;
        else hi = m;
    }
    const int32_t v1 = b.v1[r];
    const int64_t sidx = (int64_t)blockIdx.x - b.wg0[r];
    uint32_t *cnt = dense + (size_t)r * (size_t)stride;
    for (int x = threadIdx.x; x < CYC_CACHE; x += blockDim.x) {
        keys[x] = -1;
        cnts[x] = 0;
    }
    __syncthreads();
    const int64_t per_span = span / block;  // span is a multiple of block
    uint64_t acc = cyc_wave_pass<false, true>(g, wp, v1, sidx * per_span + wave, CYC_HEAVY_WAVES, cnt,
                                              s_pre + wave * GPM_WAVE, s_beg + wave * GPM_WAVE, block,
                                              (sidx + 1) * per_span, keys, cnts);
    __syncthreads();
    for (int x = threadIdx.x; x < CYC_CACHE; x += blockDim.x) {
        const uint32_t k = cnts[x];
        if (k == 0) continue;
        const uint64_t old = atomicAdd(cnt + keys[x], k);
        acc += old * k + (uint64_t)k * (k - 1) / 2;
    }
    uint64_t bytes = 0;
    if (sidx == 0 && threadIdx.x == 0) {
        const int64_t lb = g.off[0][v1], le = g.mid[0][v1];
        bytes = 8ull * (uint64_t)(wp[le] - wp[lb]) + 4ull + 12ull * (uint64_t)(le - lb);
    }
    acc = wave_sum_u64(acc);
    if (lane == 0 && acc) atomicAdd(n_cycles, (unsigned long long)acc);
    if (bytes) atomicAdd(alg_bytes, (unsigned long long)bytes);
}

// Heavy roots with counters held in LDS.  A random 4-byte atomic on a counter
// array in global memory costs a whole cache line of traffic for every cold
// counter and queues on every hot one.  Here the counters of root v1 are cut
// into ranges of S ids counted down from v1 (range q covers
// [v1 - (q + 1) S, v1 - q S) below v1, so the range next to v1 -- the highest
// degrees, the hottest counters -- is always a full one) and a workgroup of 16
// waves keeps the S counters of one range at a time in LDS.  Its task is one
// block of up to 1024 * tiles arcs v1 -> v2 below v1, one arc per lane and
// `tiles` (at most CYC_RANGE_TILES) tiles of 64 arcs per wave, and it takes the ranges from the
// top down: lists are ascending, so the two-paths v1 - v2 - v3 with v3 inside
// a range are one contiguous piece of N(v2) that ends where the piece of the
// range above began -- every lane keeps that position and finds the piece's
// start by galloping backwards from it (pieces are short: a probe or two, in
// the cache line just read).  A wave scan of the piece lengths concatenates a
// tile's 64 pieces, lanes take consecutive two-paths of the concatenation and
// add 1 to the LDS counter of v3; the old values add up to sum k(k-1)/2 over
// the range's counters.  A root with one block is done with that.  With
// several blocks every task then adds its non-zero counters k to the root's
// array in global memory (root r of the batch: dense + r * stride, zeroed by
// the caller afterwards), and an old value o there stands for o * k more
// pairs: at most S atomics per task and range instead of one per two-path,
// all inside one 4 S-byte piece of the array.  A task ends with the range in
// which its last list runs out.
// Workgroups wg0[r] .. wg0[r + 1] are the arc blocks of root r of the batch;
// with an empty batch workgroup i is the only block of root singles[i * single_step]
// (any number of roots per launch, the list lives in global memory, heaviest root
// first: a launch takes every single_step-th root of it).
// Dynamic LDS (words): cnt[S] | s_beg[16][64] (int64) | s_pre[16][64].
#define CYC_RANGE_MAX_IDS 32768
#define CYC_RANGE_TILES 4
#define CYC_RANGE_LDS_BYTES(S) ((S) * 4 + CYC_HEAVY_WAVES * GPM_WAVE * 12)

__global__ __launch_bounds__(CYC_HEAVY_WAVES *GPM_WAVE) void k_cycle4_ranges(GraphView g, const int64_t *wp, CycBatch b,
                                                                            const int32_t *singles, int single_step, int S, int tiles,
                                                                            uint32_t *dense, int64_t stride,
                                                                            unsigned long long *n_cycles,
                                                                            unsigned long long *alg_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ int32_t s_top[2];
    uint32_t *cnt = smem;
    const int wave = threadIdx.x / GPM_WAVE, lane = lane_id();
    int64_t *s_beg = reinterpret_cast<int64_t *>(smem + S) + wave * GPM_WAVE;  // S is a multiple of 4
    uint32_t *s_pre = smem + S + CYC_HEAVY_WAVES * GPM_WAVE * 2 + wave * GPM_WAVE;
    int r = 0;  // last root with wg0[r] <= blockIdx.x
    for (int hi = b.n; hi - r > 1;) {
        const int m = (r + hi) >> 1;
        if (b.wg0[m] <= (int)blockIdx.x) r = m;
        else hi = m;
    }
    const int32_t v1 = b.n ? b.v1[r] : singles[(size_t)blockIdx.x * (size_t)single_step];
    const int64_t *off = g.off[0];
    const int32_t *adj = g.adj[0];
    const int64_t lb = off[v1], le = g.mid[0][v1];
    const int blk = b.n ? (int)blockIdx.x - b.wg0[r] : 0;
    const int nblk = b.n ? b.wg0[r + 1] - b.wg0[r] : 1;
    const bool merge = nblk > 1;
    // This lane's arcs.  Neighbouring arcs (similar ids, so similar degrees: the last arcs of a hub lead
    // to the other hubs, whose lists are the longest) go to different blocks first, then to different
    // waves, then to different lanes, so that every block and every wave gets its share of the long lists.
    // lo[k] .. lo[k] + left[k] is what is left of N(v2) below the ranges done so far (at first: below v1,
    // the arc's two-path count)
    int64_t lo[CYC_RANGE_TILES];
    uint32_t left[CYC_RANGE_TILES];
#pragma unroll
    for (int k = 0; k < CYC_RANGE_TILES; ++k) {
        const int64_t a = lb + ((int64_t)((k * GPM_WAVE + lane) * CYC_HEAVY_WAVES + wave)) * nblk + blk;
        lo[k] = 0;
        left[k] = 0;
        if (k < tiles && a < le) {
            lo[k] = off[adj[a]];
            left[k] = (uint32_t)(wp[a + 1] - wp[a]);
        }
    }
    if (threadIdx.x == 0) s_top[0] = s_top[1] = -1;
    uint64_t acc = 0;
    // The last entry of every list that has entries left, all tiles requested together; the highest of them
    // over the workgroup is where a pass's range ends, so no pass is spent on ids nobody has left, and a list
    // has entries in the range exactly when its last entry is at or above id_lo.  The entries for the next
    // pass are requested as soon as a list's piece for this pass is known, so that the read is under way
    // while the piece is counted.
    int32_t last[CYC_RANGE_TILES];
#pragma unroll
    for (int k = 0; k < CYC_RANGE_TILES; ++k) last[k] = left[k] ? adj[lo[k] + left[k] - 1] : -1;
    for (int pass = 0;; ++pass) {
        __syncthreads();  // the last pass is over (walk and flush), s_top[pass & 1] is -1
        int32_t top = -1;
#pragma unroll
        for (int k = 0; k < CYC_RANGE_TILES; ++k) top = max(top, last[k]);
        for (int x = threadIdx.x; x < S / 4; x += blockDim.x) reinterpret_cast<uint4 *>(cnt)[x] = make_uint4(0u, 0u, 0u, 0u);
        for (int o = 32; o > 0; o >>= 1) top = max(top, __shfl_xor(top, o, GPM_WAVE));
        if (lane == 0 && top >= 0) atomicMax(&s_top[pass & 1], top);
        __syncthreads();
        const int32_t id_hi = s_top[pass & 1] + 1;
        if (id_hi <= 0) break;  // every list has run out
        if (threadIdx.x == 0) s_top[(pass & 1) ^ 1] = -1;  // read last before the barrier at the top of this pass
        const int32_t id_lo = max(0, id_hi - S);  // this pass counts the v3 in [id_lo, id_hi)
#pragma unroll
        for (int k = 0; k < CYC_RANGE_TILES; ++k) {
            // first entry of what is left that lies in the range: backwards in doubling steps while the
            // entry is still inside, then a binary search in the last step
            uint32_t keep = left[k];  // entries that stay for the ranges below
            if (last[k] >= id_lo) {
                const int32_t *list = adj + lo[k];
                keep -= 1;
                uint32_t step = 2;
                while (keep >= step && list[keep - step] >= id_lo) {
                    keep -= step;
                    step <<= 1;
                }
                uint32_t b0 = keep >= step ? keep - step + 1 : 0;  // list[b0 - 1] < id_lo (or b0 == 0), list[keep] >= id_lo
                while (b0 < keep) {
                    const uint32_t m = (b0 + keep) >> 1;
                    if (list[m] < id_lo) b0 = m + 1;
                    else keep = m;
                }
            }
            const uint32_t len = left[k] - keep;
            const int64_t beg = lo[k] + keep;
            left[k] = keep;
            last[k] = keep ? adj[lo[k] + keep - 1] : -1;  // for the next pass
            const uint32_t incl = wave_incl_scan_u32(len);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, GPM_WAVE - 1);
            if (total == 0) continue;
            wave_lds_sync();
            s_pre[lane] = incl - len;  // arcs without a piece share the prefix of the next arc that has one
            s_beg[lane] = beg;
            wave_lds_sync();
            for (uint32_t w0 = 0; w0 < total; w0 += 4 * GPM_WAVE) {
                int32_t v3[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t w = w0 + u * GPM_WAVE + lane;
                    v3[u] = -1;
                    if (w < total) {
                        int j = 0;  // last arc of the tile with s_pre[j] <= w
#pragma unroll
                        for (int st = GPM_WAVE / 2; st > 0; st >>= 1)
                            if (s_pre[j + st] <= w) j += st;
                        v3[u] = adj[s_beg[j] + (w - s_pre[j])];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (v3[u] >= 0) acc += atomicAdd(cnt + (v3[u] - id_lo), 1u);
            }
        }
        if (merge) {  // this block's counts meet those of the root's other blocks
            __syncthreads();
            uint32_t *shared_cnt = dense + (size_t)r * (size_t)stride + id_lo;
            for (int x = threadIdx.x; x < id_hi - id_lo; x += blockDim.x) {
                const uint32_t k = cnt[x];
                if (k) acc += (uint64_t)atomicAdd(shared_cnt + x, k) * k;
            }
        }
    }
    uint64_t bytes = 0;
    if (blk == 0 && threadIdx.x == 0) bytes = 8ull * (uint64_t)(wp[le] - wp[lb]) + 4ull + 12ull * (uint64_t)(le - lb);
    acc = wave_sum_u64(acc);
    if (lane == 0 && acc) atomicAdd(n_cycles, (unsigned long long)acc);
    if (bytes) atomicAdd(alg_bytes, (unsigned long long)bytes);
}

// ---------------------------------------------------------------------------
// Octahedra (K6 minus a perfect matching): 4-cycles among the common higher
// neighbours of a pair of opposite vertices
// ---------------------------------------------------------------------------
//
// The pattern's six vertices form three pairs of opposite (non-adjacent) vertices, every other pair is
// adjacent.  A copy is counted at the pair (u, u') that holds its lowest vertex u (ids are degree ranks): its
// other four vertices are higher neighbours of u that are neighbours of u' as well, and they form a
// 4-cycle.  With U = N+(u) and T = U ^ N(u'), the copies at (u, u') are the 4-cycles of the subgraph T
// induces: half the sum over pairs {x, y} of T of c(c-1)/2, c = |N(x) ^ N(y) ^ T|.  Each copy stands for 48
// embeddings (the reference reports every automorphic image, reference: src/gpm-par.c:179-243).
//
// Two kernels per round of roots, each root with a slab of global memory of its own for the round.
//
// k_octa_mask: one workgroup per root u.  Every vertex t above u gets a bit mask over the positions of U:
// the members of U adjacent to t.  The masks live in the root's slab (room for W + 1 words per vertex of the
// graph, rounded up to an even number, all zero between rounds; a root uses only as many words per vertex as
// its own universe needs): the waves walk the lists of U's members above u and OR the member's bit into the
// mask of every entry (atomics in L2; word Wd of a mask says which of its words hold bits, and who finds it
// empty appends the vertex to the slab's list of touched vertices).  The masks of U's own members are the
// adjacency inside U, both directions: they are copied out as rows, and a header records the universe's size
// and the number of touched vertices.
//
// k_octa_sets: the touched vertices of all roots of the round, cut into tasks of 64 or 256 vertices that
// persistent workgroups take from a queue -- a root with 10^5 touched vertices is spread over the whole GPU
// instead of keeping one workgroup busy for milliseconds.  A task copies its root's rows to LDS; every
// touched vertex t with at least four bits is a u': a wave lists the members of its mask and its lanes take
// the pairs {x, y}, three-way AND of two LDS rows and the mask.  Masks are cleared as they are read, so the
// slab is all zero again when the round ends.
// Universes of up to OCTA_MAX_D vertices; the caller checks the longest N+ and the memory for the slabs.
#define OCTA_MAX_D 512
#define OCTA_MLP 8     // mask bits set per lane and trip
#define OCTA_ROW_WORDS(W) (((W) + 3) & ~3)  // rows are read 16 bytes at a time
// dynamic LDS: k_octa_mask U[D] | list begins[D] (64-bit) | list lengths[D]; k_octa_sets rows[D * Wr] | per wave: members[D / 2] (16-bit) | sets[OCTA_SETS_WORDS(W, seg_wide)];
// D = 32 W, Wr = W rounded up to 4
#define OCTA_MASK_LDS_BYTES(W) (4 * 32 * (W) * 4)
// room for the sets of one task: SEGW wide ones (universes above 64 vertices) or 256 of up to two words
#define OCTA_SETS_WORDS(W, SEGW) ((SEGW) * ((W) + 1) > 768 ? (SEGW) * ((W) + 1) : 768)
#define OCTA_SETS_LDS_BYTES(W, SEGW) ((32 * (W) * OCTA_ROW_WORDS(W) + 4 * (32 * (W) / 2) + OCTA_SETS_WORDS(W, SEGW)) * 4)

// What k_octa_mask leaves for k_octa_sets about the root of one slab.
struct OctaHeader {
    int32_t d;       // size of the universe, 0: nothing to do
    int32_t nt;      // touched vertices
    int32_t seg;     // touched vertices per task of k_octa_sets: 64 for wide universes (large sets), else 256
    int32_t ntasks;  // ceil(nt / seg)
};

// Layout of one root's masks and rows, from the size of its universe.
struct OctaGeom {
    int Wd, Wrd, rstride;
    bool one_word, half_word;
};
__device__ __forceinline__ OctaGeom octa_geom(int d) {
    OctaGeom q;
    q.Wd = (d + 31) >> 5;
    q.Wrd = OCTA_ROW_WORDS(q.Wd);
    q.one_word = d <= 64;   // one 64-bit word per mask: the word itself tells the first setter
    q.half_word = d <= 32;  // one 32-bit word: twice as many masks to a cache line again
    // a root's masks are as wide as its universe and lie that close together in the slab: the narrower, the
    // more of them share a cache line
    q.rstride = q.half_word ? 1 : q.one_word ? 2 : (q.Wd + 2) & ~1;
    return q;
}

// grid = roots of the round (at most the slabs of the launch's group); NW waves per workgroup.
template <int NW>
__global__ __launch_bounds__(NW * GPM_WAVE) void k_octa_mask(GraphView g, const int32_t *order, int64_t lo, int64_t hi,
                                                             uint32_t *mask_all, int32_t *touched_all, uint32_t *rows_all,
                                                             OctaHeader *hdr, int W, int seg_wide,
                                                             unsigned long long *alg_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ int s_nt;
    constexpr int NT = NW * GPM_WAVE;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / GPM_WAVE;
    const int64_t ri = lo + blockIdx.x;
    if (ri >= hi) return;
    int32_t *U = (int32_t *)smem;
    const int stride = (W + 2) & ~1;  // even: the first two words of a mask are also one 64-bit word
    uint32_t *mask = mask_all + (size_t)blockIdx.x * (size_t)g.n * (size_t)stride;
    int32_t *touched = touched_all + (size_t)blockIdx.x * (size_t)g.n;
    uint32_t *rows_g = rows_all + (size_t)blockIdx.x * (size_t)(32 * W) * (size_t)OCTA_ROW_WORDS(W);
    const int64_t *off = g.off[0], *mid = g.mid[0];
    const int32_t *adj = g.adj[0];
    const int32_t u = order[ri];  // roots in the order of their estimated work, heaviest first
    const int64_t ub = mid[u];
    const int d = (int)(off[u + 1] - ub);
    if (d < 4) {
        if (tid == 0) hdr[blockIdx.x] = OctaHeader{0, 0, 64, 0};
        return;
    }
    if (tid == 0) s_nt = 0;
    // the part of every member's list above u, found by all threads at once (one member each): a wave that
    // searched for the lists of its members one after the other waited for a chain of a dozen loads per member
    int64_t *Lb = reinterpret_cast<int64_t *>(smem + 32 * W);
    int32_t *Ln = (int32_t *)(smem + 32 * W + 2 * 32 * W);
    for (int x = tid; x < d; x += NT) {
        const int32_t mvert = adj[ub + x];
        U[x] = mvert;
        const int64_t e = off[mvert + 1];
        const int64_t b = lower_bound_g(adj, off[mvert], e, u + 1);
        Lb[x] = b;
        Ln[x] = (int32_t)(e - b);
    }
    const OctaGeom q = octa_geom(d);
    const int Wd = q.Wd, rstride = q.rstride;
    uint64_t bytes = 0;  // the root, its higher neighbours and every list entry walked, 4 each
    __syncthreads();
    // masks: member x sets its bit in the mask of every neighbour above u (lists ascend: entered above u)
    for (int x = wave; x < d; x += NW) {
        const int64_t b = Lb[x], e = b + Ln[x];
        const uint32_t bit = 1u << (x & 31);
        if (lane == 0) bytes += 4ull * (uint64_t)(e - b) + 4ull + (x == 0 ? 4ull : 0ull);
        // OCTA_MLP entries per lane and trip: their atomics are all issued before the first old value is
        // looked at (an atomic that returns a value costs a round trip to the L2)
        for (int64_t p0 = b; p0 < e; p0 += OCTA_MLP * GPM_WAVE) {
            int32_t t[OCTA_MLP];
            uint32_t old[OCTA_MLP];
#pragma unroll
            for (int k = 0; k < OCTA_MLP; ++k) {
                const int64_t p = p0 + k * GPM_WAVE + lane;
                t[k] = p < e ? adj[p] : -1;
            }
            if (q.half_word) {
#pragma unroll
                for (int k = 0; k < OCTA_MLP; ++k) old[k] = t[k] >= 0 ? atomicOr(mask + (size_t)t[k], bit) : 1u;
            } else if (q.one_word) {
#pragma unroll
                for (int k = 0; k < OCTA_MLP; ++k)
                    old[k] = t[k] >= 0 ? (atomicOr(reinterpret_cast<unsigned long long *>(mask + 2 * (size_t)t[k]), 1ull << x) != 0ull) : 1u;
            } else {
#pragma unroll
                for (int k = 0; k < OCTA_MLP; ++k)
                    old[k] = t[k] >= 0 ? atomicOr(mask + (size_t)t[k] * (size_t)rstride + (x >> 5), bit) : 1u;
#pragma unroll
                for (int k = 0; k < OCTA_MLP; ++k)
                    if (old[k] == 0u) old[k] = atomicOr(mask + (size_t)t[k] * (size_t)rstride + Wd, 1u << (x >> 5));
            }
#pragma unroll
            for (int k = 0; k < OCTA_MLP; ++k)
                if (old[k] == 0u) touched[atomicAdd(&s_nt, 1)] = t[k];
        }
    }
    __syncthreads();
    // the masks of U's members are the adjacency inside U (read past the L1: they were written by atomics)
    for (int x = tid; x < d * q.Wrd; x += NT) {
        const int i = x / q.Wrd, w = x - i * q.Wrd;
        rows_g[x] = w < Wd ? __hip_atomic_load(mask + (size_t)U[i] * (size_t)rstride + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    if (tid == 0) {
        const int seg = d > 64 ? seg_wide : 256;  // the sets kernel is launched with LDS for seg_wide wide sets
        hdr[blockIdx.x] = OctaHeader{d, s_nt, seg, (s_nt + seg - 1) / seg};
    }
    bytes = wave_sum_u64(bytes);
    if (lane == 0 && bytes) atomicAdd(alg_bytes, (unsigned long long)bytes);
}

// Persistent workgroups of four waves; nslabs = roots of the round.  LDS (static): the running task counts of
// the round's roots.
__global__ __launch_bounds__(256) void k_octa_sets(GraphView g, int nslabs, const OctaHeader *hdr, uint32_t *mask_all,
                                                   const int32_t *touched_all, const uint32_t *rows_all, int W,
                                                   unsigned long long *queue, unsigned long long *n_copies) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ int s_first[1025];  // tasks of the slabs before slab i (nslabs <= 1024)
    __shared__ int s_task;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / GPM_WAVE;
    const int Dcap = 32 * W, Wr = OCTA_ROW_WORDS(W);
    uint32_t *rows = smem;
    uint16_t *members = (uint16_t *)(smem + Dcap * Wr + wave * (Dcap / 2));
    uint32_t *sets = smem + Dcap * Wr + 4 * (Dcap / 2);  // masks (and sizes) of the task's vertices with four bits or more
    __shared__ int s_nsets;
    const int stride = (W + 2) & ~1;
    {  // exclusive prefix of the task counts: every thread sums a run of four slabs, the runs are scanned
        int mine[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mine[k] = 4 * tid + k < nslabs ? hdr[4 * tid + k].ntasks : 0;
            sum += mine[k];
        }
        __shared__ int s_wave[4];
        int incl = sum;
        for (int o = 1; o < GPM_WAVE; o <<= 1) {
            const int up = __shfl_up(incl, o, GPM_WAVE);
            if (lane >= o) incl += up;
        }
        if (lane == GPM_WAVE - 1) s_wave[wave] = incl;
        __syncthreads();
        int before = incl - sum;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s_first[4 * tid + k] = before;
            before += mine[k];
        }
        if (tid == 255) s_first[1024] = before;
    }
    __syncthreads();
    const int ntasks = s_first[1024];
    // Every 4-cycle of a set T is found at both of its diagonals: the lanes add what their pairs find, and
    // the sum over the wave -- whole sets only, so an even number -- is halved at the end.
    uint64_t twice = 0;
    int have = -1;  // slab whose rows are in LDS
    for (;;) {
        __syncthreads();  // the last task is done with s_task and the LDS rows
        if (tid == 0) s_task = (int)atomicAdd(queue, 1ull);
        __syncthreads();
        const int task = s_task;
        if (task >= ntasks) break;
        int slab = 0;  // last slab whose first task is <= task and that has tasks
        for (int b = 512; b > 0; b >>= 1)
            if (slab + b < 1024 && s_first[slab + b] <= task) slab += b;
        const int seg = task - s_first[slab];
        const OctaHeader h = hdr[slab];
        const int d = h.d, nt = h.nt;
        const OctaGeom q = octa_geom(d);
        const int Wd = q.Wd, Wrd = q.Wrd, Wqd = Wrd / 4, rstride = q.rstride;
        const bool one_word = q.one_word, half_word = q.half_word;
        // rows lie an odd number of 16-byte pieces apart where that fits: piece v of rows r and r + 8 would
        // share its banks at a stride of 8 words
        const int Wsd = (Wrd | 4) * d <= Dcap * Wr ? (Wrd | 4) : Wrd;
        uint32_t *mask = mask_all + (size_t)slab * (size_t)g.n * (size_t)stride;
        const int32_t *touched = touched_all + (size_t)slab * (size_t)g.n;
        if (slab != have) {
            const uint32_t *rows_g = rows_all + (size_t)slab * (size_t)Dcap * (size_t)Wr;
            for (int x = tid; x < d * Wrd; x += 256) {
                const int i = x / Wrd, w = x - i * Wrd;
                rows[i * Wsd + w] = rows_g[x];
            }
            have = slab;
            __syncthreads();
        }
        // Most touched vertices are adjacent to fewer than four members.  Every thread takes one touched vertex
        // of the task: it reads and clears the mask of its vertex (the reads of all masks are in flight
        // together) and counts its bits; the masks with at least four bits go to a list in LDS.
        if (tid == 0) s_nsets = 0;
        __syncthreads();
        const int SW = Wd + 1;  // words of a listed set: its mask and its size
        {
            const int z = seg * h.seg + tid;
            if (tid < h.seg && z < nt) {
                const int32_t t = touched[z];
                uint32_t *mt = mask + (size_t)t * (size_t)rstride;
                // only the words that hold bits are read (one, for most vertices)
                const uint32_t filled = half_word ? 1u : one_word ? 3u : __hip_atomic_load(mt + Wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t mw[OCTA_MAX_D / 32];
                int k = 0;
#pragma unroll
                for (int w = 0; w < OCTA_MAX_D / 32; ++w)
                    mw[w] = (filled >> w) & 1u ? __hip_atomic_load(mt + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
                for (int w = 0; w < OCTA_MAX_D / 32; ++w) {
                    k += __popc(mw[w]);
                    if (mw[w]) atomicExch(mt + w, 0u);  // cleared for the next round
                }
                if (!one_word) atomicExch(mt + Wd, 0u);
                if (k >= 4) {
                    uint32_t *slot = sets + atomicAdd(&s_nsets, 1) * SW;
#pragma unroll
                    for (int w = 0; w < OCTA_MAX_D / 32; ++w)
                        if (w < Wd) slot[w] = mw[w];
                    slot[Wd] = (uint32_t)k;
                }
            }
        }
        __syncthreads();
        // The listed sets are dealt out to the waves in turn (the first touched vertices of a heavy root are
        // its hubs, with the largest sets: neighbours in the list go to different waves).
        const int nsets = s_nsets;
        for (int si = wave; si < nsets; si += 4) {
            const uint32_t *slot = sets + si * SW;
            // the set's mask, the same in every lane (words from Wd on are zeros)
            uint32_t um[OCTA_MAX_D / 32];
#pragma unroll
            for (int w = 0; w < OCTA_MAX_D / 32; ++w) um[w] = w < Wd ? __builtin_amdgcn_readfirstlane(slot[w]) : 0u;
            const int kk = (int)__builtin_amdgcn_readfirstlane(slot[Wd]);
            wave_lds_sync();  // the last set's pairs are done with members[]
            {  // positions of the mask's bits, in ascending order: lane l lists the bits of word l
                int at = 0;
                uint32_t bits = 0;
#pragma unroll
                for (int w = 0; w < OCTA_MAX_D / 32; ++w) {
                    if (lane > w) at += __popc(um[w]);
                    if (lane == w) bits = um[w];
                }
                while (bits) {
                    members[at++] = (uint16_t)(32 * lane + __builtin_ctz(bits));
                    bits &= bits - 1;
                }
            }
            wave_lds_sync();
            // The kk (kk - 1) / 2 pairs are dealt out to the lanes as one sequence (row p holds the pairs
            // (p, p + 1) .. (p, kk - 1)): most sets are small, and a lane per pair of one row at a time
            // would leave most lanes idle.  A lane keeps its place as (p, r) = r-th pair of row p.
            int pp = 0, r = lane;
            while (pp < kk - 1 && r >= kk - 1 - pp) {
                r -= kk - 1 - pp;
                ++pp;
            }
            while (pp < kk - 1) {
                const uint32_t *rp = rows + (int)members[pp] * Wsd, *rq = rows + (int)members[pp + 1 + r] * Wsd;
                uint32_t c = 0;
#pragma unroll
                for (int v = 0; v < OCTA_MAX_D / 128; ++v)
                    if (v < Wqd)
                        c = and_popc_add4(and4(load4(rp + 4 * v), u32x4{um[4 * v], um[4 * v + 1], um[4 * v + 2], um[4 * v + 3]}),
                                          load4(rq + 4 * v), c);
                twice += (uint64_t)c * (c - (c ? 1u : 0u)) / 2;
                r += GPM_WAVE;
                while (pp < kk - 1 && r >= kk - 1 - pp) {
                    r -= kk - 1 - pp;
                    ++pp;
                }
            }
        }
    }
    twice = wave_sum_u64(twice);
    if (lane == 0 && twice) atomicAdd(n_copies, (unsigned long long)(twice / 2));
}

// Work estimate of the octahedron kernel for every root of [lo, hi) (root index i is vertex n-1-i): the
// entries its members' lists hold above the root -- one mask bit each -- weighted by the width of the universe,
// which bounds the size of the sets whose pairs are tested.  keys[i - lo] = estimate, verts[i - lo] = vertex.
__global__ void k_octa_cost(GraphView g, int64_t lo, int64_t hi, uint64_t *keys, int32_t *verts) {
    const int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    const int32_t u = (int32_t)(g.n - 1 - i);
    const int64_t ub = g.mid[0][u], ue = g.off[0][u + 1];
    uint64_t bits = 0;
    if (ue - ub >= 4)
        for (int64_t p = ub; p < ue; ++p) {
            const int32_t x = g.adj[0][p];
            const int64_t e = g.off[0][x + 1];
            bits += (uint64_t)(e - lower_bound_g(g.adj[0], g.off[0][x], e, u + 1));
        }
    keys[i - lo] = bits * (uint64_t)(8 + (ue - ub)) + 1;
    verts[i - lo] = u;
}

// Longest N+ of the graph (the universes of the clique and octahedron kernels).
__global__ void k_max_plus(GraphView g, int *out) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int d = v < g.n ? (int)(g.off[0][v + 1] - g.mid[0][v]) : 0;
    for (int o = 32; o > 0; o >>= 1) d = max(d, __shfl_xor(d, o, GPM_WAVE));
    if (lane_id() == 0 && d > 0) atomicMax(out, d);
}

// ---------------------------------------------------------------------------
// Output: matches as text lines, formatted on the GPU
// ---------------------------------------------------------------------------
//
// The reference prints one line per match, "m(1) m(2) ... m(P)\n" with single
// spaces (reference: writeMatch, src/gpm-par.c:71-77), one fprintf per number
// on rank 0.  Here a canonical match row (pattern-vertex order, the caller's
// 1-based ids) becomes n_aut lines, one per automorphic image; line lengths
// are computed first, an exclusive scan places the lines, and a second kernel
// writes the characters.  One thread per line.

__device__ __forceinline__ int dec_digits(uint32_t x) {
    return x < 10u ? 1 : x < 100u ? 2 : x < 1000u ? 3 : x < 10000u ? 4 : x < 100000u ? 5 : x < 1000000u ? 6
         : x < 10000000u ? 7 : x < 100000000u ? 8 : x < 1000000000u ? 9 : 10;
}

// aut[a * P + v] = which column of the canonical row image a prints at place v.
__global__ void k_line_len(const int32_t *rows, int64_t nrows, int P, int n_aut, const int8_t *aut, int64_t *len) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * n_aut) return;
    const int32_t *m = rows + (idx / n_aut) * P;
    int bytes = P;  // P - 1 spaces and the newline
    for (int v = 0; v < P; ++v) bytes += dec_digits((uint32_t)m[v]);  // every image prints the same P ids
    len[idx] = bytes;
}

__global__ void k_line_write(const int32_t *rows, int64_t nrows, int P, int n_aut, const int8_t *aut,
                             const int64_t *off, char *text) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * n_aut) return;
    const int32_t *m = rows + (idx / n_aut) * P;
    const int8_t *im = aut + (idx % n_aut) * P;
    char *p = text + off[idx];
    for (int v = 0; v < P; ++v) {
        uint32_t x = (uint32_t)m[n_aut > 1 ? im[v] : v];
        const int nd = dec_digits(x);
        for (int k = nd - 1; k >= 0; --k) {
            p[k] = (char)('0' + x % 10u);
            x /= 10u;
        }
        p += nd;
        *p++ = v + 1 < P ? ' ' : '\n';
    }
}
