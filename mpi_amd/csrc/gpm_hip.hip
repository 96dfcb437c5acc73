// libgpm_hip.so -- host side of the MI355X match search: the C ABI declared in
// include/gpm_hip.h, graph preparation in HBM, and the chunked level-by-level
// driver that launches the kernels of gpm_kernels.hpp.
#include <cstring>  // before rocprim: its texture iterator calls host memset
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gpm_hip.h"
#include "gpm_kernels.hpp"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? GPM_ENOMEM : GPM_ERUNTIME, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                    \
    } while (0)

extern "C" const char *gpm_hip_last_error(void) { return g_err.c_str(); }

// ---------------------------------------------------------------------------
// fixed device memory of the entry point that allocates nothing
// ---------------------------------------------------------------------------
//
// gpm_hip_count_noalloc() keeps its frontiers, candidate ranges, prefix sums
// and counters in this array instead of allocated buffers.  Every GPU the
// library runs on holds its own copy until the process ends.  On the host the
// array's name is only a handle: the address a kernel can use comes from
// hipGetSymbolAddress, once per device, and is what every launch, memset and
// copy below is given.
#define NOALLOC_ROWS ((int64_t)1 << 16)      // rows per frontier level
#define NOALLOC_BYTES ((size_t)32 << 20)     // 32 MiB per GPU
#define NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) unsigned char gpm_scratch[NOALLOC_BYTES];
static unsigned char *g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};
static std::mutex g_scratch_mu;

// Address of this device's copy of gpm_scratch; `dev` must be the current device.
static int scratch_address(int dev, unsigned char **out) {
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return fail(GPM_EINVAL, "device %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        unsigned char *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(gpm_scratch));
        if (e != hipSuccess || !ptr)
            return fail(GPM_ENODEV, "hipGetSymbolAddress(gpm_scratch) on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return GPM_OK;
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------

enum {  // slots of the per-device counter block (unsigned long long each)
    CTR_LEVEL0 = 0,   // [0, GPM_MAXP): rows produced into level k's frontier
    CTR_MATCHES = 16, // matches counted by the level-extension kernel
    CTR_BYTES = 17,   // merge-model bytes summed by k_ranges
    CTR_TAIL_MATCHES = 18,
    CTR_TAIL_BYTES = 19,
    CTR_EMIT = 20,
    CTR_CYC_BYTES = 21,       // merge-model bytes of the 4-cycle kernels
    CTR_OCTA_BYTES = 22,      // bytes of the lists the octahedron kernel walks
    CTR_OVER = 30,            // [30, 33): partial matches handed back by the launches (tiers) of the three-level kernel
    CTR_KIND_BYTES = 40,      // [40, 40 + GPM_KERNEL_KINDS): bytes summed out of the clique slots, per kind
    CTR_CLQ_CLASS0 = 24,  // [24, 29): roots per size class of the clique kernel
    CTR_COUNT = 48
};

struct EventPair {
    hipEvent_t a, b;
    int kind;  // GPM_KIND_*
    int64_t rows;
    int tag;   // size class of a clique launch (for GPM_TRACE), else 0
};

static const char *const kKindNames[GPM_KERNEL_KINDS] = {"k_tail", "k_extend", "k_clique", "k_clq_build",
                                                         "k_clq_count4", "k_cycle4", "k_octa", ""};

// One chunk of the root range; in clique mode also its roots per size class.
struct Chunk {
    int64_t lo, hi;
    uint32_t n[CLQ_CLASSES];
};

// The clique path keeps several chunks in flight, each on its own stream with
// its own root lists, so that the tail of one chunk's launches overlaps the
// next chunk's.
#define CLQ_INFLIGHT 4
#define CLQ_ROWS_WORDS ((size_t)1 << 29)  // rows buffer per in-flight chunk: 2 GiB
struct ChunkSlot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    int32_t *lists = nullptr;
    unsigned long long *counts = nullptr;
    uint32_t *scratch = nullptr;  // bitmap rows of the roots too large for LDS, one slab per workgroup
    uint32_t *rows = nullptr;     // rows buffer of the build/count kernel pair
    bool busy = false;
};

struct DevState {
    int dev = -1;
    hipStream_t stream = nullptr;
    // graph
    int32_t n = 0;
    int64_t m[2] = {0, 0};
    int64_t *off[2] = {nullptr, nullptr}, *mid[2] = {nullptr, nullptr};
    int32_t *adj[2] = {nullptr, nullptr};
    int32_t *perm = nullptr;
    uint8_t *label = nullptr;
    uint32_t *hub = nullptr, *hubc = nullptr;  // hub matrix and its summary (one bit per word)
    uint32_t *labcnt = nullptr, *labseg = nullptr;
    int32_t *adjL = nullptr;
    GraphView gv{};
    // search buffers
    int64_t cap_rows = 0, cap_emit = 0;
    int32_t *F[GPM_MAXP] = {};
    int64_t *beg[GPM_MAXP] = {}, *cnt[GPM_MAXP] = {}, *pref[GPM_MAXP] = {};
    int8_t *wsel[GPM_MAXP] = {};  // which operand list each row walks at the next level
    int64_t *over_rows = nullptr;  // partial matches the three-level kernel hands back
    bool no_fuse3 = false;         // ... while they go through the level-by-level path
    void *scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    int32_t *emit = nullptr, *emit_host = nullptr;
    // output mode on several GPUs: rows wait here for the next gather round (see gather_round)
    int32_t *ag_send = nullptr, *ag_recv = nullptr;  // cap_emit rows; ngpus x cap_emit rows
    unsigned long long *ag_counts = nullptr;         // [0] this GPU's row count, [1, 1 + ngpus) everybody's
    uint64_t ag_rows = 0;                            // rows waiting in ag_send
    int index = 0;                                   // position of this device in the context
    // text output: automorphism table, line offsets, text on the device and two pinned host buffers
    int8_t *aut_dev = nullptr;
    int64_t *line_off = nullptr;
    char *text_dev = nullptr, *text_host[2] = {nullptr, nullptr};
    int64_t text_lines_cap = 0, text_line_max = 0;
    void *text_scan_tmp = nullptr;
    size_t text_scan_bytes = 0;
    unsigned long long *ctr = nullptr;       // device counters
    unsigned long long *ctr_host = nullptr;  // pinned mirror for read-back
    int64_t *i64_dev = nullptr, *i64_host = nullptr;
    // clique kernel: per-class root lists, global row scratch, spread result slots
    ChunkSlot slots[CLQ_INFLIGHT];
    int64_t clq_stride = 0;
    int clq_scratch_blocks = 0;
    int clq_probe_bias = 0;    // GPM_CLIQUE_PROBE_BIAS, read once per context
    bool count4_wide = true;   // GPM_COUNT4_WAVES >= 16
    // GPM_COUNT4_VT_<class 2..5>: columns per tile of the count (256, 512, 1024).  Tiles of 512 leave out a quarter
    // of the words a universe of 1024 would AND and let two workgroups share a compute unit; tiles of 256 list
    // every row's bits so often that the listing costs more than the ANDs saved.
    int count4_vt[4] = {512, 512, 512, 1024};
    int count4_wide_min = 512;  // GPM_COUNT4_WIDE_MIN: tiles of at least this many columns are counted by 16 waves
    // Diagnostics that give wrong counts (a warning says so): GPM_CLQ_PROBE, bits: 1 the list kernel reads its lists and
    // tests nothing, 2 the same with half of every list, 4 no hub-matrix gathers, 8 no rows written;
    // GPM_COUNT4_PROBE bits: 1 the count kernel lists and ANDs nothing, 2 it writes no list entries either.
    int clq_probe = 0, count4_probe = 0;
    int clq_lists_max_class = 5;  // GPM_CLQ_LISTS_MAX_CLASS: size classes up to this one build their rows from the
                                  // adjacency lists, larger ones from hub-matrix gathers (1 = gathers only)
    int clq_build_rb = 0;         // GPM_CLQ_BUILD_RB: workgroups per root of that kernel (0 = by size class)
    int clq_lists_waves = 16;     // GPM_CLQ_LISTS_WAVES: waves per workgroup of that kernel (4 or 16)
    // GPM_CLQ_KAPPA4 (all classes) / GPM_CLQ_KAPPA4_<class 2..5>: a row of that kernel whose list has this many quarter
    // entries per position left to test is gathered from the hub matrix instead (-1: never)
    int clq_kappa4[4] = {6, 3, 3, 3};  // measured per class on the headline graph
    unsigned long long *clq_slots = nullptr;  // [0,CLQ_SLOTS) matches, [CLQ_SLOTS,2*CLQ_SLOTS) bytes
    // 4-cycle path: two-path count of every root, counter arrays (cyc_blocks of them for the workgroups of
    // the light kernel, then CYC_HEAVY_GROUPS groups of cyc_batch for heavy roots), root queues of the light
    // kernel's launches
    int64_t *cyc_cost = nullptr;
    int64_t *cyc_wp = nullptr;  // two-paths of all arcs before arc p (m + 1 entries)
    uint32_t *cyc_dense = nullptr;
    int64_t cyc_stride = 0;
    int cyc_blocks = 0;
    int cyc_batch = 1;     // heavy roots per launch
    int32_t *cyc_single = nullptr;  // copy of gpm_ctx::cyc_single_v1
    uint64_t cyc_single_ver = 0;
    LeafTerm *ls_terms = nullptr;  // inclusion-exclusion terms of the leaf-suffix kernel
    unsigned long long *cyc_queue = nullptr;
    // octahedron path: per-workgroup slabs of masks ((octa_w + 1) words per vertex) and touched lists, root queues
    uint32_t *octa_mask = nullptr;
    int32_t *octa_touched = nullptr;
    uint32_t *octa_rows = nullptr;   // per slab: the adjacency inside the root's universe, as k_octa_mask leaves it
    OctaHeader *octa_hdr = nullptr;  // per slab: size of the universe, touched vertices, tasks
    unsigned long long *octa_queue = nullptr;
    int octa_wgs = 0, octa_w = 0;  // octa_wgs = -1: the graph does not fit the kernel (longest N+, memory)
    int32_t *octa_order = nullptr;  // copy of gpm_ctx::octa_order (roots, heaviest first)
    uint64_t octa_order_ver = 0;
    std::vector<EventPair> events;
    size_t events_used = 0;
    // 4-cycle path: completion of the last CLQ_INFLIGHT chunks, one event per slot stream and chunk, so that a
    // worker is never more than that many chunks ahead of its GPU
    std::vector<hipEvent_t> ring;
    // results of the last search
    uint64_t canonical = 0, alg_bytes = 0, tail_bytes = 0, chunks = 0;
    uint64_t rows_in[GPM_MAXP] = {};
    double ms_kind[GPM_KERNEL_KINDS] = {};
    uint64_t launches_kind[GPM_KERNEL_KINDS] = {}, bytes_kind[GPM_KERNEL_KINDS] = {};
    int rc = 0;
    std::string err;
    // true for the view gpm_hip_count_noalloc() searches with: every buffer lies in gpm_scratch, nothing is
    // (re)allocated and launches are not timed (events would have to be created)
    bool fixed = false;
};

struct RowBatch {
    std::vector<int32_t> rows;
    std::vector<char> text;  // text mode: whole lines
    uint64_t nrows;
};

struct gpm_ctx {
    std::vector<DevState> devs;
    gpm_graph_info info{};
    bool have_graph = false, have_plan = false;
    gpm_plan plan{};
    std::vector<int8_t> aut;
    PlanView pv{};
    TailInfo ti{};
    Tail3Info t3{};
    LeafSuffix ls{};
    bool use_tail = true;
    int clique_k = 0;  // K when the plan is a K-clique under a total ordering, else 0
    bool use_clique = true;
    bool cycle4 = false;  // the plan is the 4-cycle with all 8 automorphisms folded
    bool use_cycle4 = true;
    bool octa = false;    // the plan is K6 minus a perfect matching with all 48 automorphisms folded
    bool use_octa = true;
    bool accounting = true;  // sum merge-model bytes in the clique kernels (gpm_hip_set_accounting)
    // tuning knobs, read from the environment once when the context is made
    int64_t fuse3_group = 8, twin_group = 1;
    bool fuse3_wave = true, fuse3_mid = true, trace = false;
    int clique_max_d_env = -1;   // GPM_CLIQUE_MAX_D, -1 = unset
    int clique_small_max = 256;  // GPM_CLIQUE_SMALL_MAX
    int clique_tiny = 32;        // GPM_CLIQUE_TINY: largest universe of the lane-group kernel (0, 16, 32 or 64)
    int64_t cyc_heavy_block = 2048, cyc_span = (int64_t)1 << 18;
    CycRule cyc_rule{};                     // which 4-cycle kernel takes a root (thresholds, counters per LDS range)
    int cyc_range_tiles = CYC_RANGE_TILES;  // tiles of 64 arcs per wave of such a task
    std::vector<int64_t> cyc_heavy_roots;  // root indices at or above the threshold, ascending
    std::vector<int64_t> cyc_heavy_paths;  // their two-path counts
    std::vector<int64_t> cyc_heavy_arcs;   // their arcs to lower vertices
    // heavy roots that count in LDS ranges and whose arcs fit one task: root indices (ascending) and
    // vertices; every device keeps a copy of the vertex list (cyc_single), refreshed when the version moves
    std::vector<int64_t> cyc_single_roots;
    std::vector<int32_t> cyc_single_v1;
    uint64_t cyc_single_ver = 0;
    int64_t chunk_roots = 1 << 20;  // upper bound of a chunk's size
    int64_t chunk_hint = 0;         // how many chunks to aim for (0 = 32 per GPU of the context)
    int64_t min_chunk_cost = (int64_t)1 << 22;  // GPM_MIN_CHUNK_COST: least estimated work of a chunk, in the units of
                                                // k_root_cost / two-paths (tests set 1 to cut tiny graphs into many chunks)
    std::vector<Chunk> chunks;
    bool chunks_valid = false;
    // octahedron path: the roots sorted by estimated work, heaviest first, and the chunk table over that order
    // (lo, hi are positions in octa_order); empty when the kernel cannot take the graph
    std::vector<int32_t> octa_order;
    std::vector<Chunk> octa_chunks;
    uint64_t octa_order_ver = 0;
    int octa_streams = 2;  // GPM_OCTA_STREAMS: launches in flight, each on its own stream with its own share of the slabs
    int octa_seg = 64;  // GPM_OCTA_SEG: touched vertices per task of k_octa_sets for roots with more than 64 higher neighbours
    int64_t octa_wide_roots = 0;  // GPM_OCTA_WIDE: chunks of at most this many roots build their masks with 16 waves per root (measured: no gain)
    int64_t root_lo = 0, root_hi = 0;
    volatile uint64_t *shared_cursor = nullptr;
    std::atomic<uint64_t> own_cursor{0};
    // row hand-off from device threads to the calling thread
    std::mutex mu;
    std::condition_variable cv;
    std::deque<RowBatch> queue;
    int workers_live = 0;
    std::atomic<bool> stop{false};
    gpm_stats stats{};
    std::vector<void *> comms;  // one RCCL communicator per GPU (several GPUs, or GPM_FORCE_RCCL=1)
    // gather rounds of an output-mode search on several GPUs (see gather_round)
    bool gather_on_device = true;     // GPM_GATHER=host: every worker hands its own rows to the host queue instead
    struct {
        bool round_failed = false;  // a worker could not prepare its part of the current round
        int arrived = 0, finished = 0, waiting = 0;
        uint64_t generation = 0, wait_generation = 0;
        std::atomic<bool> requested{false};
        uint64_t rows[2][8] = {};     // per round parity: rows every worker brings
        bool last[2] = {false, false};  // per round parity: every worker had finished when the round formed
        uint64_t rounds = 0, gathered_rows = 0;
    } gs;
    DevState na;                      // view of devs[0] whose search buffers lie in gpm_scratch
    unsigned char *scratch0 = nullptr;  // devs[0]'s copy of gpm_scratch
};

// ---------------------------------------------------------------------------
// RCCL, bound at run time: only a context that drives several GPUs needs it
// ---------------------------------------------------------------------------
//
// The per-GPU match counts of one context are summed with one ncclAllReduce
// over xGMI (the reference gathers results with one MPI message per match,
// reference: src/gpm-par.c:181, :608-621).  The library is opened with dlopen
// so that single-GPU users carry no RCCL dependency.

struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm,
                     hipStream_t stream) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return lib != nullptr; }
};
enum { RCCL_SUM = 0, RCCL_INT32 = 2, RCCL_UINT64 = 5 };  // ncclSum, ncclInt32, ncclUint64 (rccl.h)

static Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *lib = nullptr;
        for (const char *n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) return;
        Rccl t;
        t.CommInitAll = (decltype(t.CommInitAll))dlsym(lib, "ncclCommInitAll");
        t.CommDestroy = (decltype(t.CommDestroy))dlsym(lib, "ncclCommDestroy");
        t.AllReduce = (decltype(t.AllReduce))dlsym(lib, "ncclAllReduce");
        t.AllGather = (decltype(t.AllGather))dlsym(lib, "ncclAllGather");
        t.GroupStart = (decltype(t.GroupStart))dlsym(lib, "ncclGroupStart");
        t.GroupEnd = (decltype(t.GroupEnd))dlsym(lib, "ncclGroupEnd");
        t.GetErrorString = (decltype(t.GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (t.CommInitAll && t.CommDestroy && t.AllReduce && t.AllGather && t.GroupStart && t.GroupEnd &&
            t.GetErrorString) {
            t.lib = lib;
            r = t;
        }
    });
    return r;
}

static int64_t env_i64(const char *name, int64_t dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoll(v) : dflt;
}

static double now_seconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// GPM_PROGRESS=<seconds>: a line on stderr at most that often while a long search runs.
static void progress_tick(int dev, uint64_t chunk, size_t n_chunks, int level, int64_t rows) {
    static const int64_t every = env_i64("GPM_PROGRESS", 0);
    static std::atomic<double> last{0.0};
    if (every <= 0) return;
    const double t = now_seconds();
    if (t - last.load() < (double)every) return;
    last.store(t);
    fprintf(stderr, "gpm progress: device %d chunk %llu of %zu, level %d, %lld partial matches in hand\n", dev,
            (unsigned long long)chunk, n_chunks, level, (long long)rows);
}

extern "C" int gpm_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        fail(GPM_ENODEV, "hipGetDeviceCount failed");
        return 0;
    }
    return n;
}

// Null-safe frees: freeing a null pointer would leave a sticky HIP error that
// the next library call (rocprim checks hipGetLastError) would trip over.
static void dev_free(void *p) {
    if (p) (void)hipFree(p);
}
static void host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

// Device memory that lives for one call: released on every return path, error returns included.
struct Scratch {
    void *p = nullptr;
    Scratch() = default;
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    ~Scratch() { dev_free(p); }
    hipError_t alloc(size_t bytes) {
        dev_free(p);
        p = nullptr;
        return hipMalloc(&p, bytes ? bytes : 8);
    }
    void release() {
        dev_free(p);
        p = nullptr;
    }
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

static void free_graph(DevState &d) {
    (void)hipSetDevice(d.dev);
    if (d.off[1] == d.off[0]) {  // symmetric graph: the in-CSR is an alias, not a second allocation
        d.off[1] = nullptr;
        d.adj[1] = nullptr;
        d.mid[1] = nullptr;
    }
    for (int dir = 0; dir < 2; ++dir) {
        dev_free(d.off[dir]);
        dev_free(d.adj[dir]);
        dev_free(d.mid[dir]);
        d.off[dir] = nullptr;
        d.adj[dir] = nullptr;
        d.mid[dir] = nullptr;
    }
    dev_free(d.perm);
    dev_free(d.label);
    dev_free(d.hub);
    dev_free(d.hubc);
    dev_free(d.labcnt);
    dev_free(d.labseg);
    dev_free(d.adjL);
    d.labseg = nullptr;
    d.adjL = nullptr;
    dev_free(d.cyc_cost);
    dev_free(d.cyc_wp);
    d.cyc_wp = nullptr;
    dev_free(d.cyc_dense);
    dev_free(d.cyc_queue);
    dev_free(d.cyc_single);
    d.cyc_single = nullptr;
    d.cyc_single_ver = 0;
    dev_free(d.octa_mask);
    dev_free(d.octa_touched);
    dev_free(d.octa_queue);
    dev_free(d.octa_rows);
    dev_free(d.octa_hdr);
    d.octa_rows = nullptr;
    d.octa_hdr = nullptr;
    d.octa_mask = nullptr;
    d.octa_touched = nullptr;
    d.octa_queue = nullptr;
    d.octa_wgs = d.octa_w = 0;
    dev_free(d.octa_order);
    d.octa_order = nullptr;
    d.octa_order_ver = 0;
    d.cyc_cost = nullptr;
    d.cyc_dense = nullptr;
    d.cyc_queue = nullptr;
    d.labcnt = nullptr;
    d.perm = nullptr;
    d.label = nullptr;
    d.hub = d.hubc = nullptr;
}

static void free_buffers(DevState &d) {
    (void)hipSetDevice(d.dev);
    for (int k = 0; k < GPM_MAXP; ++k) {
        dev_free(d.F[k]);
        dev_free(d.beg[k]);
        dev_free(d.cnt[k]);
        dev_free(d.pref[k]);
        dev_free(d.wsel[k]);
        d.wsel[k] = nullptr;
        d.F[k] = nullptr;
        d.beg[k] = d.cnt[k] = d.pref[k] = nullptr;
    }
    dev_free(d.scan_tmp);
    dev_free(d.over_rows);
    d.over_rows = nullptr;
    for (auto &sl : d.slots) {
        dev_free(sl.lists);
        dev_free(sl.counts);
        dev_free(sl.scratch);
        dev_free(sl.rows);
        sl.rows = nullptr;
        sl.lists = nullptr;
        sl.counts = nullptr;
        sl.scratch = nullptr;
    }
    dev_free(d.clq_slots);
    d.clq_slots = nullptr;
    dev_free(d.aut_dev);
    dev_free(d.line_off);
    dev_free(d.text_dev);
    dev_free(d.text_scan_tmp);
    d.text_scan_tmp = nullptr;
    host_free(d.text_host[0]);
    host_free(d.text_host[1]);
    d.aut_dev = nullptr;
    d.line_off = nullptr;
    d.text_dev = nullptr;
    d.text_host[0] = d.text_host[1] = nullptr;
    d.text_lines_cap = 0;
    dev_free(d.emit);
    host_free(d.emit_host);
    dev_free(d.ag_send);
    dev_free(d.ag_recv);
    dev_free(d.ag_counts);
    d.ag_send = d.ag_recv = nullptr;
    d.ag_counts = nullptr;
    d.scan_tmp = nullptr;
    d.emit = nullptr;
    d.emit_host = nullptr;
}

extern "C" void gpm_hip_destroy(gpm_ctx *ctx) {
    if (!ctx) return;
    for (void *c : ctx->comms)
        if (c) (void)rccl().CommDestroy(c);
    for (auto &d : ctx->devs) {
        (void)hipSetDevice(d.dev);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        free_graph(d);
        free_buffers(d);
        dev_free(d.ls_terms);
        d.ls_terms = nullptr;
        for (auto &ev : d.events) {
            (void)hipEventDestroy(ev.a);
            (void)hipEventDestroy(ev.b);
        }
        for (auto &ev : d.ring) (void)hipEventDestroy(ev);
        for (auto &sl : d.slots) {
            if (sl.done) (void)hipEventDestroy(sl.done);
            if (sl.stream) (void)hipStreamDestroy(sl.stream);
        }
        dev_free(d.ctr);
        host_free(d.ctr_host);
        dev_free(d.i64_dev);
        host_free(d.i64_host);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    delete ctx;
}

extern "C" int gpm_hip_create(int ngpus, const int *devices, gpm_ctx **out) {
    if (!out || ngpus < 1 || ngpus > 8) return fail(GPM_EINVAL, "ngpus must be 1..8");
    int have = gpm_hip_device_count();
    if (have < 1) return fail(GPM_ENODEV, "no HIP device is visible");
    gpm_ctx *ctx = new gpm_ctx();
    ctx->devs.resize((size_t)ngpus);
    for (int i = 0; i < ngpus; ++i) {
        DevState &d = ctx->devs[(size_t)i];
        d.index = i;
        d.dev = devices ? devices[i] : i;
        if (d.dev < 0 || d.dev >= have) {
            gpm_hip_destroy(ctx);
            return fail(GPM_EINVAL, "device %d not present (%d visible)", d.dev, have);
        }
        hipError_t e = hipSetDevice(d.dev);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&d.ctr, CTR_COUNT * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipHostMalloc(&d.ctr_host, CTR_COUNT * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc(&d.i64_dev, 4 * sizeof(int64_t));
        if (e == hipSuccess) e = hipHostMalloc(&d.i64_host, 4 * sizeof(int64_t));
        for (auto &sl : d.slots) {
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            gpm_hip_destroy(ctx);
            return fail(GPM_ENODEV, "set-up of device %d failed: %s", d.dev, hipGetErrorString(e));
        }
        // this device's copy of the fixed scratch array (device d.dev is current)
        unsigned char *sp = nullptr;
        int src = scratch_address(d.dev, &sp);
        if (src != GPM_OK) {
            std::string why = g_err;
            gpm_hip_destroy(ctx);
            return fail(src, "%s", why.c_str());
        }
        if (i == 0) ctx->scratch0 = sp;
    }
    ctx->chunk_roots = env_i64("GPM_CHUNK_ROOTS", 1 << 20);
    if (ctx->chunk_roots < 1) ctx->chunk_roots = 1;
    ctx->chunk_hint = env_i64("GPM_CHUNKS", 0);
    ctx->min_chunk_cost = std::max<int64_t>(1, env_i64("GPM_MIN_CHUNK_COST", (int64_t)1 << 22));
    ctx->use_tail = env_i64("GPM_NO_TAIL", 0) == 0;
    ctx->use_clique = env_i64("GPM_NO_CLIQUE", 0) == 0;
    ctx->use_cycle4 = env_i64("GPM_NO_CYCLE4", 0) == 0;
    ctx->use_octa = env_i64("GPM_NO_OCTA", 0) == 0;
    ctx->octa_streams = (int)std::min<int64_t>(CLQ_INFLIGHT, std::max<int64_t>(1, env_i64("GPM_OCTA_STREAMS", 2)));
    ctx->octa_wide_roots = env_i64("GPM_OCTA_WIDE", 0);
    {
        const int64_t sg = env_i64("GPM_OCTA_SEG", 64);
        ctx->octa_seg = sg >= 256 ? 256 : sg >= 128 ? 128 : 64;
    }
    ctx->cyc_rule.heavy_from = std::max<int64_t>(1, env_i64("GPM_CYCLE4_HEAVY", (int64_t)1 << 16));
    ctx->accounting = env_i64("GPM_ACCOUNTING", 1) != 0;
    {
        const char *gm = getenv("GPM_GATHER");
        ctx->gather_on_device = !(gm && strcmp(gm, "host") == 0);
    }
    ctx->fuse3_group = env_i64("GPM_FUSE3_GROUP", 8);
    ctx->fuse3_wave = env_i64("GPM_FUSE3_WAVE", 1) != 0;
    ctx->fuse3_mid = env_i64("GPM_FUSE3_MID", 1) != 0;
    ctx->twin_group = env_i64("GPM_TWIN_GROUP", 1);
    ctx->trace = getenv("GPM_TRACE") != nullptr;
    ctx->clique_max_d_env = (int)env_i64("GPM_CLIQUE_MAX_D", -1);
    ctx->clique_small_max = (int)std::min<int64_t>(256, std::max<int64_t>(4, env_i64("GPM_CLIQUE_SMALL_MAX", 256)));
    {
        const int64_t t = env_i64("GPM_CLIQUE_TINY", 32);
        ctx->clique_tiny = t >= 64 ? 64 : t >= 32 ? 32 : t >= 16 ? 16 : 0;
    }
    // GPM_CYC_RANGES=0: no root counts in LDS ranges (heavy roots are cut into spans on global counter arrays)
    ctx->cyc_rule.S = env_i64("GPM_CYC_RANGES", 1) == 0
                          ? 0
                          : (int)std::min<int64_t>(CYC_RANGE_MAX_IDS, std::max<int64_t>(256, env_i64("GPM_CYC_RANGE_IDS", 16384))) / 4 * 4;
    ctx->cyc_rule.ranged_from = std::max<int64_t>(1, env_i64("GPM_CYC_RANGES_FROM", (int64_t)1 << 12));
    ctx->cyc_rule.per_range = (int32_t)std::max<int64_t>(0, env_i64("GPM_CYC_RANGE_WORK", 64));
    ctx->cyc_range_tiles = (int)std::min<int64_t>(CYC_RANGE_TILES, std::max<int64_t>(1, env_i64("GPM_CYC_RANGE_TILES", CYC_RANGE_TILES)));
    ctx->cyc_heavy_block = std::max<int64_t>(256, env_i64("GPM_CYC_HEAVY_BLOCK", 2048)) / 256 * 256;
    ctx->cyc_span = std::max<int64_t>(1, env_i64("GPM_CYC_SPAN", (int64_t)1 << 18) / ctx->cyc_heavy_block) * ctx->cyc_heavy_block;
    for (auto &d : ctx->devs) {
        d.clq_probe_bias = (int)env_i64("GPM_CLIQUE_PROBE_BIAS", 0);
        d.count4_wide = env_i64("GPM_COUNT4_WAVES", 16) >= 16;
        for (int c = 2; c <= 5; ++c) {
            char name[32];
            snprintf(name, sizeof name, "GPM_COUNT4_VT_%d", c);
            const int64_t vt = env_i64(name, d.count4_vt[c - 2]);
            d.count4_vt[c - 2] = vt <= 256 ? 256 : vt <= 512 ? 512 : 1024;
        }
        d.count4_wide_min = (int)env_i64("GPM_COUNT4_WIDE_MIN", d.count4_wide_min);
        d.clq_probe = (int)env_i64("GPM_CLQ_PROBE", 0);
        d.count4_probe = (int)env_i64("GPM_COUNT4_PROBE", 0);
        if (d.clq_probe || d.count4_probe) {
            static std::atomic<bool> warned{false};
            if (!warned.exchange(true)) fprintf(stderr, "gpm: GPM_CLQ_PROBE / GPM_COUNT4_PROBE set: clique counts are wrong (diagnostic run)\n");
        }
        d.clq_lists_max_class = (int)env_i64("GPM_CLQ_LISTS_MAX_CLASS", 5);
        d.clq_build_rb = (int)env_i64("GPM_CLQ_BUILD_RB", 0);
        d.clq_lists_waves = env_i64("GPM_CLQ_LISTS_WAVES", 16) == 4 ? 4 : 16;
        for (int c = 2; c <= 5; ++c) {
            char name[32];
            snprintf(name, sizeof name, "GPM_CLQ_KAPPA4_%d", c);
            d.clq_kappa4[c - 2] = (int)env_i64(name, env_i64("GPM_CLQ_KAPPA4", d.clq_kappa4[c - 2]));
        }
    }
    // GPM_NO_RCCL=1: per-GPU counts are summed on the host only (also lets two workers share one GPU in tests)
    if ((ngpus > 1 && env_i64("GPM_NO_RCCL", 0) == 0) || env_i64("GPM_FORCE_RCCL", 0)) {
        if (!rccl().ok()) {
            gpm_hip_destroy(ctx);
            return fail(GPM_ENODEV, "a multi-GPU context needs librccl.so, which could not be loaded");
        }
        std::vector<int> devlist;
        for (auto &d : ctx->devs) devlist.push_back(d.dev);
        ctx->comms.assign((size_t)ngpus, nullptr);
        int rc = rccl().CommInitAll(ctx->comms.data(), ngpus, devlist.data());
        if (rc != 0) {
            std::string why = rccl().GetErrorString(rc);
            ctx->comms.clear();
            gpm_hip_destroy(ctx);
            return fail(GPM_ENODEV, "ncclCommInitAll failed: %s", why.c_str());
        }
    }
    *out = ctx;
    return GPM_OK;
}

// ---------------------------------------------------------------------------
// graph upload: copy, validate, renumber by degree, rebuild sorted CSR
// ---------------------------------------------------------------------------

static inline unsigned grid_for(int64_t n, int block = 256) { return (unsigned)((n + block - 1) / block); }

static int bits_for(int64_t n) {
    int b = 1;
    while (((int64_t)1 << b) < n) ++b;
    return b;
}

// Build one direction of the renumbered CSR on the current device.
static int rebuild_direction(DevState &d, int dir, const int64_t *old_off, const int32_t *old_adj, int64_t m,
                             const int32_t *rank, uint32_t *flags) {
    const int32_t n = d.n;
    hipStream_t s = d.stream;
    Scratch new_deg, rank_tmp, tmp, keys, keys2;
    HIP_TRY(hipMalloc(&d.off[dir], ((size_t)n + 1) * sizeof(int64_t)));
    HIP_TRY(hipMalloc(&d.mid[dir], ((size_t)n + 1) * sizeof(int64_t)));
    HIP_TRY(hipMalloc(&d.adj[dir], ((size_t)m + 1) * sizeof(int32_t)));
    HIP_TRY(new_deg.alloc(((size_t)n + 1) * sizeof(int64_t)));
    HIP_TRY(rank_tmp.alloc(((size_t)n + 1) * sizeof(int32_t)));  // k_scatter_rank writes rank again; harmless, keeps one kernel
    if (n > 0) k_scatter_rank<<<grid_for(n), 256, 0, s>>>(n, d.perm, rank_tmp.as<int32_t>(), old_off, new_deg.as<int64_t>());
    HIP_TRY(hipMemsetAsync(d.off[dir], 0, sizeof(int64_t), s));
    size_t tmp_bytes = 0;
    if (n > 0) {
        HIP_TRY(rocprim::inclusive_scan(nullptr, tmp_bytes, new_deg.as<int64_t>(), d.off[dir] + 1, (size_t)n,
                                        rocprim::plus<int64_t>(), s));
        HIP_TRY(tmp.alloc(tmp_bytes));
        HIP_TRY(rocprim::inclusive_scan(tmp.p, tmp_bytes, new_deg.as<int64_t>(), d.off[dir] + 1, (size_t)n,
                                        rocprim::plus<int64_t>(), s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (m > 0) {
        HIP_TRY(keys.alloc((size_t)m * sizeof(uint64_t)));
        HIP_TRY(keys2.alloc((size_t)m * sizeof(uint64_t)));
        k_arc_keys<<<grid_for(m), 256, 0, s>>>(n, m, old_off, old_adj, rank, keys.as<uint64_t>());
        unsigned end_bit = 32u + (unsigned)bits_for(n);
        tmp_bytes = 0;
        HIP_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, keys.as<uint64_t>(), keys2.as<uint64_t>(), (size_t)m, 0u,
                                         end_bit, s));
        HIP_TRY(tmp.alloc(tmp_bytes));
        HIP_TRY(rocprim::radix_sort_keys(tmp.p, tmp_bytes, keys.as<uint64_t>(), keys2.as<uint64_t>(), (size_t)m, 0u,
                                         end_bit, s));
        k_keys_to_adj<<<grid_for(m), 256, 0, s>>>(m, keys2.as<uint64_t>(), d.adj[dir], flags);
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (n > 0) k_mid<<<grid_for(n), 256, 0, s>>>(n, d.off[dir], d.adj[dir], d.mid[dir]);
    HIP_TRY(hipStreamSynchronize(s));
    d.m[dir] = m;
    return GPM_OK;
}

static int upload_one(DevState &d, int32_t n, const int64_t *out_off, const int32_t *out_adj, const int64_t *in_off,
                      const int32_t *in_adj, const uint8_t *labels, uint32_t *flags_out) {
    HIP_TRY(hipSetDevice(d.dev));
    free_graph(d);
    hipStream_t s = d.stream;
    d.n = n;
    const int64_t m_out = out_off[n];
    const int64_t m_in = in_off ? in_off[n] : 0;
    Scratch o_off[2], o_adj[2], flags_buf;
    HIP_TRY(flags_buf.alloc(sizeof(uint32_t)));
    uint32_t *flags = flags_buf.as<uint32_t>();
    HIP_TRY(hipMemsetAsync(flags, 0, sizeof(uint32_t), s));
    const int ndir = in_off ? 2 : 1;
    for (int dir = 0; dir < ndir; ++dir) {
        const int64_t *ho = dir ? in_off : out_off;
        const int32_t *ha = dir ? in_adj : out_adj;
        int64_t m = dir ? m_in : m_out;
        HIP_TRY(o_off[dir].alloc(((size_t)n + 1) * sizeof(int64_t)));
        HIP_TRY(o_adj[dir].alloc(((size_t)m + 1) * sizeof(int32_t)));
        HIP_TRY(hipMemcpyAsync(o_off[dir].p, ho, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
        if (m > 0) HIP_TRY(hipMemcpyAsync(o_adj[dir].p, ha, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, s));
        k_validate_off<<<grid_for((int64_t)n + 1), 256, 0, s>>>(n, m, o_off[dir].as<int64_t>(), flags);
        if (m > 0) k_validate_adj<<<grid_for(m), 256, 0, s>>>(n, m, o_adj[dir].as<int32_t>(), flags);
    }
    uint32_t hflags = 0;
    HIP_TRY(hipMemcpyAsync(&hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (hflags & 2u)
        return fail(GPM_EINVAL, "CSR arrays are inconsistent (offsets not monotone or ids out of range)");

    // renumber by ascending total degree, ties by id (stable radix sort)
    Scratch deg, deg2, ids, rank;
    HIP_TRY(deg.alloc(((size_t)n + 1) * sizeof(uint32_t)));
    HIP_TRY(deg2.alloc(((size_t)n + 1) * sizeof(uint32_t)));
    HIP_TRY(ids.alloc(((size_t)n + 1) * sizeof(int32_t)));
    HIP_TRY(rank.alloc(((size_t)n + 1) * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&d.perm, ((size_t)n + 1) * sizeof(int32_t)));
    if (n > 0) {
        k_total_degree<<<grid_for(n), 256, 0, s>>>(n, o_off[0].as<int64_t>(), o_off[1].as<int64_t>(), deg.as<uint32_t>(),
                                                   ids.as<int32_t>());
        size_t tmp_bytes = 0;
        Scratch tmp, scratch_deg;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, deg.as<uint32_t>(), deg2.as<uint32_t>(), ids.as<int32_t>(),
                                          d.perm, (size_t)n, 0u, 32u, s));
        HIP_TRY(tmp.alloc(tmp_bytes));
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, deg.as<uint32_t>(), deg2.as<uint32_t>(), ids.as<int32_t>(),
                                          d.perm, (size_t)n, 0u, 32u, s));
        HIP_TRY(scratch_deg.alloc(((size_t)n + 1) * sizeof(int64_t)));
        k_scatter_rank<<<grid_for(n), 256, 0, s>>>(n, d.perm, rank.as<int32_t>(), o_off[0].as<int64_t>(),
                                                   scratch_deg.as<int64_t>());
        HIP_TRY(hipStreamSynchronize(s));
    }
    int rc = rebuild_direction(d, 0, o_off[0].as<int64_t>(), o_adj[0].as<int32_t>(), m_out, rank.as<int32_t>(), flags);
    if (rc == GPM_OK && in_off)
        rc = rebuild_direction(d, 1, o_off[1].as<int64_t>(), o_adj[1].as<int32_t>(), m_in, rank.as<int32_t>(), flags);
    if (rc != GPM_OK) return rc;
    if (!in_off) {
        d.off[1] = d.off[0];
        d.adj[1] = d.adj[0];
        d.mid[1] = d.mid[0];
        d.m[1] = d.m[0];
    }
    if (labels) {
        Scratch raw;
        HIP_TRY(raw.alloc((size_t)n + 1));
        HIP_TRY(hipMalloc(&d.label, (size_t)n + 1));
        HIP_TRY(hipMemcpyAsync(raw.p, labels, (size_t)n, hipMemcpyHostToDevice, s));
        if (n > 0) k_permute_labels<<<grid_for(n), 256, 0, s>>>(n, d.perm, raw.as<uint8_t>(), d.label);
        HIP_TRY(hipStreamSynchronize(s));
    }
    HIP_TRY(hipMemcpyAsync(&hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *flags_out = hflags;
    for (int dir = 0; dir < 2; ++dir) {  // the caller's copies are no longer needed: make room for what follows
        o_off[dir].release();
        o_adj[dir].release();
    }
    deg.release();
    deg2.release();
    ids.release();
    rank.release();
    flags_buf.release();
    d.gv.n = n;
    for (int dir = 0; dir < 2; ++dir) {
        d.gv.off[dir] = d.off[dir];
        d.gv.adj[dir] = d.adj[dir];
        d.gv.mid[dir] = d.mid[dir];
    }
    d.gv.label = d.label;

    // hub matrix over the highest-degree vertices (the last ids after renumbering)
    int64_t hub_rows = std::min<int64_t>(env_i64("GPM_HUB_ROWS", (int64_t)1 << 19), ((int64_t)n + 63) / 64 * 64);
    hub_rows = std::max<int64_t>(0, hub_rows / 64 * 64);
    d.gv.hub_base = n;
    d.gv.hub_wpr = 0;
    d.gv.hub = nullptr;
    d.gv.hub_cwpr = 0;
    d.gv.hub_dense = 0;
    d.gv.hubc = nullptr;
    if (hub_rows > 0 && n > 0) {
        const size_t bytes = (size_t)hub_rows * (size_t)(hub_rows / 32) * sizeof(uint32_t);
        // Every kernel runs without the matrix (membership by binary search, rows from hashed list walks), so a
        // device too full for it is not an error: halve the rows until it fits, or go without.
        size_t want = bytes;
        for (;;) {  // a small matrix (a small graph, a low GPM_HUB_ROWS) is tried like any other: only retries halve
            if (hipMalloc(&d.hub, want) == hipSuccess) break;
            (void)hipGetLastError();
            d.hub = nullptr;
            if (hub_rows < 2048) break;
            hub_rows = hub_rows / 2 / 64 * 64;
            want = (size_t)hub_rows * (size_t)(hub_rows / 32) * sizeof(uint32_t);
        }
        if (d.hub) {
            HIP_TRY(hipMemsetAsync(d.hub, 0, want, s));
            d.gv.hub_base = (int32_t)std::max<int64_t>(0, (int64_t)n - hub_rows);
            d.gv.hub_wpr = (int32_t)(hub_rows / 32);
            d.gv.hub = d.hub;
            // the summary (1/32 of the matrix) is optional as well: GPM_HUB_SUMMARY=0 or no memory for it
            d.gv.hub_cwpr = (d.gv.hub_wpr + 31) / 32;
            const size_t sum_bytes = (size_t)hub_rows * (size_t)d.gv.hub_cwpr * sizeof(uint32_t);
            if (env_i64("GPM_HUB_SUMMARY", 1) != 0 && hipMalloc(&d.hubc, sum_bytes) == hipSuccess) {
                HIP_TRY(hipMemsetAsync(d.hubc, 0, sum_bytes, s));
                d.gv.hubc = d.hubc;
                // the last GPM_HUB_DENSE columns (highest degrees) are probed without the summary
                d.gv.hub_dense = (int32_t)std::max<int64_t>(0, hub_rows - std::max<int64_t>(0, env_i64("GPM_HUB_DENSE", 8192)));
            } else {
                (void)hipGetLastError();
                d.hubc = nullptr;
            }
            k_hub_fill<<<(unsigned)(n - d.gv.hub_base), 256, 0, s>>>(d.gv, d.hub, d.hubc);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));
        }
    }

    // per-vertex label histograms of the neighbours (symmetric labelled graphs)
    d.gv.nlab = 0;
    d.gv.labcnt = nullptr;
    if (labels && !in_off && n > 0) {
        int nlab = 0;
        for (int32_t v = 0; v < n; ++v) nlab = std::max(nlab, (int)labels[v] + 1);
        const size_t bytes = (size_t)n * (size_t)nlab * sizeof(uint32_t);
        if (bytes <= ((size_t)2 << 30)) {
            HIP_TRY(hipMalloc(&d.labcnt, bytes));
            HIP_TRY(hipMemsetAsync(d.labcnt, 0, bytes, s));
            d.gv.nlab = nlab;
            k_label_counts<<<grid_for(n), 256, 0, s>>>(d.gv, d.labcnt);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));
            d.gv.labcnt = d.labcnt;
        }
    }
    // label-grouped copy of the lists: labelled levels read one segment instead of filtering the list
    d.gv.adjL = nullptr;
    d.gv.labseg = nullptr;
    const int64_t m_arcs = d.m[0];
    if (d.gv.labcnt && m_arcs > 0 && env_i64("GPM_NO_LABEL_SEGMENTS", 0) == 0 &&
        (size_t)n * (size_t)(d.gv.nlab + 1) * sizeof(uint32_t) <= ((size_t)4 << 30)) {
        Scratch keys, keys2, vals, tmp;
        size_t tmp_bytes = 0;
        HIP_TRY(hipMalloc(&d.labseg, (size_t)n * (size_t)(d.gv.nlab + 1) * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&d.adjL, ((size_t)m_arcs + 1) * sizeof(int32_t)));
        HIP_TRY(keys.alloc((size_t)m_arcs * sizeof(uint64_t)));
        HIP_TRY(keys2.alloc((size_t)m_arcs * sizeof(uint64_t)));
        HIP_TRY(vals.alloc((size_t)m_arcs * sizeof(int32_t)));
        k_label_segments<<<grid_for(n), 256, 0, s>>>(d.gv, d.labcnt, d.labseg);
        k_segment_keys<<<grid_for(m_arcs), 256, 0, s>>>(d.gv, m_arcs, keys.as<uint64_t>(), vals.as<int32_t>());
        HIP_TRY(hipGetLastError());
        const unsigned end_bit = 8u + (unsigned)bits_for(n);
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.as<uint64_t>(), keys2.as<uint64_t>(), vals.as<int32_t>(),
                                          d.adjL, (size_t)m_arcs, 0u, end_bit, s));
        HIP_TRY(tmp.alloc(tmp_bytes));
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys.as<uint64_t>(), keys2.as<uint64_t>(), vals.as<int32_t>(),
                                          d.adjL, (size_t)m_arcs, 0u, end_bit, s));
        HIP_TRY(hipStreamSynchronize(s));
        d.gv.adjL = d.adjL;
        d.gv.labseg = d.labseg;
    }
    return GPM_OK;
}

extern "C" int gpm_hip_upload_graph(gpm_ctx *ctx, int32_t n, const int64_t *out_off, const int32_t *out_adj,
                                    const int64_t *in_off, const int32_t *in_adj, const uint8_t *labels) {
    if (!ctx || n < 0 || !out_off || (!out_adj && out_off[n] > 0)) return fail(GPM_EINVAL, "bad graph arguments");
    (void)hipGetLastError();
    if ((in_off == nullptr) != (in_adj == nullptr) && in_off && in_off[n] > 0)
        return fail(GPM_EINVAL, "in_off and in_adj must be given together");
    if (out_off[0] != 0 || out_off[n] < 0 || (in_off && in_off[n] != out_off[n]))
        return fail(GPM_EINVAL, "arc counts of the two directions differ");
    ctx->have_graph = false;
    ctx->have_plan = false;
    ctx->chunks_valid = false;
    uint32_t flags = 0;
    if (ctx->devs.size() == 1) {
        int rc = upload_one(ctx->devs[0], n, out_off, out_adj, in_off, in_adj, labels, &flags);
        if (rc != GPM_OK) return rc;
    } else {  // every GPU copies and prepares its replica at the same time
        std::vector<std::thread> threads;
        std::vector<uint32_t> dev_flags(ctx->devs.size(), 0);
        for (size_t i = 0; i < ctx->devs.size(); ++i)
            threads.emplace_back([&, i] {
                DevState &d = ctx->devs[i];
                d.rc = upload_one(d, n, out_off, out_adj, in_off, in_adj, labels, &dev_flags[i]);
                if (d.rc) d.err = gpm_hip_last_error();
            });
        for (auto &t : threads) t.join();
        for (auto &d : ctx->devs)
            if (d.rc) return fail(d.rc, "%s", d.err.c_str());
        flags = dev_flags[0];
    }
    int64_t max_out = 0, max_in = 0;
    for (int32_t v = 0; v < n; ++v) {
        max_out = std::max(max_out, out_off[v + 1] - out_off[v]);
        if (in_off) max_in = std::max(max_in, in_off[v + 1] - in_off[v]);
    }
    ctx->info.n = n;
    ctx->info.m_out = out_off[n];
    ctx->info.symmetric = in_off == nullptr;
    ctx->info.simple = (flags & 1u) == 0;
    ctx->info.max_out_deg = (int32_t)max_out;
    ctx->info.max_in_deg = (int32_t)(in_off ? max_in : max_out);
    ctx->info.has_labels = labels != nullptr;
    ctx->info.hub_rows = ctx->devs[0].gv.hub ? ctx->devs[0].gv.hub_wpr * 32 : 0;
    ctx->root_lo = 0;
    ctx->root_hi = n;
    ctx->have_graph = true;
    return GPM_OK;
}

extern "C" int gpm_hip_graph_info(const gpm_ctx *ctx, gpm_graph_info *info) {
    if (!ctx || !info) return fail(GPM_EINVAL, "null argument");
    if (!ctx->have_graph) return fail(GPM_ESTATE, "no graph uploaded");
    *info = ctx->info;
    return GPM_OK;
}

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------

static bool same_list_set(const std::vector<std::pair<int, int>> &a, const std::vector<std::pair<int, int>> &b) {
    if (a.size() != b.size()) return false;
    for (auto &x : a) {
        bool found = false;
        for (auto &y : b) found |= x == y;
        if (!found) return false;
    }
    return true;
}

extern "C" int gpm_hip_set_plan(gpm_ctx *ctx, const gpm_plan *p) {
    if (!ctx || !p) return fail(GPM_EINVAL, "null argument");
    if (!ctx->have_graph) return fail(GPM_ESTATE, "upload the graph before the plan");
    const int P = p->n_levels;
    if (P < 1 || P > GPM_MAX_PATTERN) return fail(GPM_EINVAL, "n_levels must be 1..%d", GPM_MAX_PATTERN);
    bool seen[GPM_MAXP] = {};
    bool any_bound = false;
    for (int k = 0; k < P; ++k) {
        if (p->order[k] < 0 || p->order[k] >= P || seen[p->order[k]])
            return fail(GPM_EINVAL, "order is not a permutation of the pattern vertices");
        seen[p->order[k]] = true;
        if (p->label[k] > 255) return fail(GPM_EINVAL, "labels must fit 8 bits");
        if (p->label[k] >= 0 && !ctx->info.has_labels)
            return fail(GPM_EINVAL, "plan has labels but the graph has none");
        if (k == 0) continue;
        if (p->src_level[k] < 0 || p->src_level[k] >= k || (p->src_dir[k] | 1) != 1)
            return fail(GPM_EINVAL, "level %d: bad source", k);
        if (p->n_cons[k] < 0 || p->n_cons[k] > GPM_MAX_CONS || p->n_lower[k] < 0 || p->n_lower[k] > GPM_MAX_CONS ||
            p->n_upper[k] < 0 || p->n_upper[k] > GPM_MAX_CONS)
            return fail(GPM_EINVAL, "level %d: bad list sizes", k);
        for (int j = 0; j < p->n_cons[k]; ++j)
            if (p->cons_level[k][j] < 0 || p->cons_level[k][j] >= k || (p->cons_dir[k][j] | 1) != 1)
                return fail(GPM_EINVAL, "level %d: bad constraint %d", k, j);
        for (int j = 0; j < p->n_lower[k]; ++j)
            if (p->lower_level[k][j] < 0 || p->lower_level[k][j] >= k)
                return fail(GPM_EINVAL, "level %d: bad lower bound %d", k, j);
        for (int j = 0; j < p->n_upper[k]; ++j)
            if (p->upper_level[k][j] < 0 || p->upper_level[k][j] >= k)
                return fail(GPM_EINVAL, "level %d: bad upper bound %d", k, j);
        any_bound |= p->n_lower[k] > 0 || p->n_upper[k] > 0;
    }
    if (p->n_aut < 1 || (p->n_aut > 1 && !p->aut)) return fail(GPM_EINVAL, "bad automorphism table");
    if ((any_bound || p->n_aut > 1) && !ctx->info.simple)
        return fail(GPM_EINVAL, "ordering bounds need a data graph without duplicate arcs and self loops");

    ctx->chunks_valid = false;
    ctx->plan = *p;
    ctx->aut.assign(p->aut ? p->aut : nullptr, p->aut ? p->aut + p->n_aut * (uint64_t)P : nullptr);
    ctx->plan.aut = ctx->aut.empty() ? nullptr : ctx->aut.data();

    // order relation between levels implied by the bounds (transitive closure)
    bool lt[GPM_MAXP][GPM_MAXP] = {};  // lt[a][b]: m[a] < m[b]
    for (int k = 1; k < P; ++k) {
        for (int j = 0; j < p->n_lower[k]; ++j) lt[p->lower_level[k][j]][k] = true;
        for (int j = 0; j < p->n_upper[k]; ++j) lt[k][p->upper_level[k][j]] = true;
    }
    for (int mIdx = 0; mIdx < P; ++mIdx)
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) lt[a][b] = lt[a][b] || (lt[a][mIdx] && lt[mIdx][b]);
    auto part_of = [&](int owner, int k) { return lt[owner][k] ? PART_ABOVE : (lt[k][owner] ? PART_BELOW : PART_FULL); };

    PlanView &pv = ctx->pv;
    memset(&pv, 0, sizeof pv);
    pv.P = P;
    pv.simple = ctx->info.simple && env_i64("GPM_NO_SHORTEST", 0) == 0;
    for (int k = 0; k < P; ++k) {
        pv.order[k] = (int8_t)p->order[k];
        LevelPlan &lv = pv.lv[k];
        lv.label = p->label[k];
        lv.min_out = ctx->info.simple ? p->min_out_deg[k] : 0;
        lv.min_in = ctx->info.simple ? p->min_in_deg[k] : 0;
        if (k == 0) continue;
        lv.src_level = (int8_t)p->src_level[k];
        lv.src_dir = (int8_t)(ctx->info.symmetric ? 0 : p->src_dir[k]);
        lv.src_part = (int8_t)part_of(lv.src_level, k);
        lv.n_cons = (int8_t)p->n_cons[k];
        lv.n_lower = (int8_t)p->n_lower[k];
        lv.n_upper = (int8_t)p->n_upper[k];
        for (int j = 0; j < lv.n_cons; ++j) {
            lv.cons_level[j] = (int8_t)p->cons_level[k][j];
            lv.cons_dir[j] = (int8_t)(ctx->info.symmetric ? 0 : p->cons_dir[k][j]);
            lv.cons_part[j] = (int8_t)part_of(lv.cons_level[j], k);
        }
        for (int j = 0; j < lv.n_lower; ++j) lv.lower_level[j] = (int8_t)p->lower_level[k][j];
        for (int j = 0; j < lv.n_upper; ++j) lv.upper_level[j] = (int8_t)p->upper_level[k][j];
        lv.lower_is_src = lv.n_lower == 1 && lv.lower_level[0] == lv.src_level;
    }

    // pattern adjacency between levels, for the closed-form leaf counts
    if (ctx->info.symmetric) {
        uint16_t adjl[GPM_MAXP] = {};
        for (int k = 1; k < P; ++k) {
            const LevelPlan &lv = pv.lv[k];
            adjl[k] |= (uint16_t)(1u << lv.src_level);
            adjl[lv.src_level] |= (uint16_t)(1u << k);
            for (int j = 0; j < lv.n_cons; ++j) {
                adjl[k] |= (uint16_t)(1u << lv.cons_level[j]);
                adjl[lv.cons_level[j]] |= (uint16_t)(1u << k);
            }
        }
        for (int k = 1; k < P; ++k) pv.lv[k].src_adj = adjl[pv.lv[k].src_level];
    }
    // relation between the last two levels (fused intersection kernel)
    TailInfo &ti = ctx->ti;
    memset(&ti, 0, sizeof ti);
    ti.any_shortest = ctx->info.simple;
    if (P >= 3) {
        const int L = P - 2;
        const LevelPlan &A = pv.lv[L], &B = pv.lv[L + 1];
        std::vector<std::pair<int, int>> la, lb_rest;
        la.push_back({A.src_level, A.src_dir});
        for (int j = 0; j < A.n_cons; ++j) la.push_back({A.cons_level[j], A.cons_dir[j]});
        int n_extra = 0;
        auto addB = [&](int lvl, int dir) {
            if (lvl == L) {
                if (n_extra < 2) {
                    ti.extra_dir[n_extra] = (int8_t)dir;
                    ti.extra_part[n_extra] = (int8_t)part_of(L, L + 1);
                }
                n_extra++;
            } else {
                lb_rest.push_back({lvl, dir});
            }
        };
        addB(B.src_level, B.src_dir);
        for (int j = 0; j < B.n_cons; ++j) addB(B.cons_level[j], B.cons_dir[j]);
        bool same_bounds = true;
        for (int t = 0; t < L; ++t) same_bounds &= lt[t][L] == lt[t][L + 1] && lt[L][t] == lt[L + 1][t];
        bool same_filters = A.label == B.label && A.min_out == B.min_out && A.min_in == B.min_in;
        ti.n_extra = (int8_t)std::min(n_extra, 2);
        ti.b_lower_is_a = lt[L][L + 1];
        ti.b_upper_is_a = lt[L + 1][L];
        ti.reuse = ctx->info.simple && n_extra >= 1 && n_extra <= 2 && same_list_set(la, lb_rest) && same_bounds &&
                   same_filters && env_i64("GPM_NO_REUSE", 0) == 0;
        ti.twin = ctx->info.simple && n_extra == 0 && same_list_set(la, lb_rest) && same_bounds && same_filters &&
                  env_i64("GPM_NO_TWIN", 0) == 0;
        ti.indep = ctx->info.simple && n_extra == 0 && !ti.twin && env_i64("GPM_NO_INDEP", 0) == 0;
        // three fused levels: level Z = P-3, then twins that need Z's lists, the Z vertex's own list and
        // possibly lists of further earlier vertices, with no ordering bound between Z and the twins
        Tail3Info &t3 = ctx->t3;
        memset(&t3, 0, sizeof t3);
        if (P >= 4 && ti.twin && ctx->info.symmetric && env_i64("GPM_NO_FUSE3", 0) == 0) {
            const int Z = P - 3;
            const LevelPlan &LZ = pv.lv[Z];
            unsigned zset = 1u << LZ.src_level, aset = 1u << A.src_level;
            for (int j = 0; j < LZ.n_cons; ++j) zset |= 1u << LZ.cons_level[j];
            for (int j = 0; j < A.n_cons; ++j) aset |= 1u << A.cons_level[j];
            const bool related = lt[Z][L] || lt[L][Z];
            if ((zset & ~aset) == 0 && (aset >> Z & 1u) && !related) {
                for (int t = 0; t < Z; ++t) {
                    if (zset >> t & 1u) t3.zl[t3.n_zl++] = (int8_t)t;
                    else if (aset >> t & 1u) t3.xl[t3.n_xl++] = (int8_t)t;
                }
                t3.enabled = 1;
            }
        }
        // leaf levels: adjacency to the source is the only requirement besides the label
        const bool closed_ok = ctx->info.simple && ctx->info.symmetric && env_i64("GPM_NO_LEAF", 0) == 0;
        auto leaf = [&](const LevelPlan &X) {
            return X.n_cons == 0 && X.n_lower == 0 && X.n_upper == 0 && X.min_out <= 1 && X.min_in <= 1 &&
                   (X.label < 0 || ctx->devs[0].gv.labcnt != nullptr);
        };
        ti.leaf_b = closed_ok && leaf(B);
        const bool same_set = A.src_level == B.src_level && A.label == B.label;
        const bool disjoint = A.label >= 0 && B.label >= 0 && A.label != B.label;
        ti.leaf_pair = closed_ok && leaf(A) && leaf(B) && B.src_level != L && (same_set || disjoint);
        ti.pair_same_set = same_set;
    }
    // leaf suffix: every level from S on is a leaf (see k_leaf_suffix)
    LeafSuffix &ls = ctx->ls;
    memset(&ls, 0, sizeof ls);
    if (P >= 3 && ctx->info.simple && ctx->info.symmetric && env_i64("GPM_NO_LEAF", 0) == 0 &&
        env_i64("GPM_NO_LEAF_SUFFIX", 0) == 0) {
        auto leaf = [&](const LevelPlan &X) {
            return X.n_cons == 0 && X.n_upper == 0 && X.min_out <= 1 && X.min_in <= 1 &&
                   (X.label < 0 || ctx->devs[0].gv.labcnt != nullptr);
        };
        int S = P;
        while (S - 1 >= 1 && leaf(pv.lv[S - 1])) --S;
        bool ok = P - S >= 2;
        for (int k = S; k < P && ok; ++k) ok = pv.lv[k].src_level < S;  // leaves hang off inner vertices
        // all leaves labelled or none
        for (int k = S; k < P && ok; ++k) ok = (pv.lv[k].label < 0) == (pv.lv[S].label < 0);
        for (int k = S; k < P && ok; ++k) {
            const LevelPlan &lv = pv.lv[k];
            int c = -1;
            for (int i = 0; i < ls.n_cls; ++i)
                if (ls.cls[i].label == lv.label) c = i;
            if (c < 0) {
                if (ls.n_cls == 4) { ok = false; break; }
                c = ls.n_cls++;
                ls.cls[c].label = lv.label;
            }
            LeafClass &lc = ls.cls[c];
            int gi = -1;
            for (int i = 0; i < lc.n_grp; ++i)
                if (lc.src[i] == lv.src_level) gi = i;
            if (gi < 0) {
                if (lc.n_grp == 3) { ok = false; break; }
                gi = lc.n_grp++;
                lc.src[gi] = lv.src_level;
                lc.level0[gi] = (int8_t)k;
            }
            lc.cnt[gi]++;
            ls.cls_of[k] = (int8_t)c;
            ls.grp_of[k] = (int8_t)gi;
        }
        // ordering bounds: none, or a chain through the leaves of one group; nothing else
        for (int c = 0; c < ls.n_cls && ok; ++c)
            for (int gi = 0; gi < ls.cls[c].n_grp && ok; ++gi) {
                int prev = -1, chained = 0, members = 0;
                for (int k = S; k < P && ok; ++k) {
                    if (ls.cls_of[k] != c || ls.grp_of[k] != gi) continue;
                    const LevelPlan &lv = pv.lv[k];
                    members++;
                    if (lv.n_lower == 0) {
                        ok = chained == 0;  // either every later member is chained or none is
                    } else {
                        ok = lv.n_lower == 1 && lv.lower_level[0] == prev;
                        chained++;
                    }
                    prev = k;
                }
                ok = ok && (chained == 0 || chained == members - 1);
                ls.cls[c].ordered[gi] = chained > 0;
            }
        // bounds from leaves towards other levels would show up as lower bounds of those levels
        for (int k = 1; k < P && ok; ++k)
            for (int j = 0; j < pv.lv[k].n_lower && ok; ++j)
                if (pv.lv[k].lower_level[j] >= S) ok = k >= S && ls.cls_of[k] == ls.cls_of[pv.lv[k].lower_level[j]] &&
                                                       ls.grp_of[k] == ls.grp_of[pv.lv[k].lower_level[j]];
        // classes of three groups: the terms of the inclusion-exclusion over the set partitions of their leaves
        std::vector<LeafTerm> terms;
        for (int c = 0; c < ls.n_cls && ok; ++c) {
            LeafClass &lc = ls.cls[c];
            if (lc.n_grp < 3) continue;
            std::vector<int> grp;  // group of every leaf of the class
            for (int gi = 0; gi < 3; ++gi) grp.insert(grp.end(), (size_t)lc.cnt[gi], gi);
            const int q = (int)grp.size();
            std::map<std::array<uint8_t, 8>, long long> sum;
            std::vector<int> blk((size_t)q, 0);  // restricted growth string: block of every leaf
            for (;;) {
                int nb = 0;
                for (int b : blk) nb = std::max(nb, b + 1);
                int mask[GPM_MAXP] = {0}, size[GPM_MAXP] = {0};
                for (int i = 0; i < q; ++i) {
                    mask[blk[(size_t)i]] |= 1 << grp[(size_t)i];
                    size[blk[(size_t)i]]++;
                }
                long long coef = 1;
                std::array<uint8_t, 8> e{};
                for (int b = 0; b < nb; ++b) {
                    long long f = 1;
                    for (int i = 2; i < size[b]; ++i) f *= i;  // (size - 1)!
                    coef *= (size[b] % 2 ? f : -f);
                    e[(size_t)mask[b]]++;
                }
                sum[e] += coef;
                // next restricted growth string
                int i = q - 1;
                for (; i > 0; --i) {
                    int mx = 0;
                    for (int j = 0; j < i; ++j) mx = std::max(mx, blk[(size_t)j]);
                    if (blk[(size_t)i] <= mx) break;
                }
                if (i == 0) break;
                blk[(size_t)i]++;
                for (int j = i + 1; j < q; ++j) blk[(size_t)j] = 0;
            }
            lc.term_beg = (int16_t)terms.size();
            for (const auto &kv : sum)
                if (kv.second) {
                    LeafTerm t{};
                    t.coef = kv.second;
                    for (int k = 0; k < 8; ++k) t.e[k] = kv.first[(size_t)k];
                    terms.push_back(t);
                }
            lc.term_end = (int16_t)terms.size();
            ok = terms.size() < 30000;
        }
        for (auto &d : ctx->devs) {
            HIP_TRY(hipSetDevice(d.dev));
            dev_free(d.ls_terms);
            d.ls_terms = nullptr;
            if (ok && !terms.empty()) {
                HIP_TRY(hipMalloc(&d.ls_terms, terms.size() * sizeof(LeafTerm)));
                HIP_TRY(hipMemcpy(d.ls_terms, terms.data(), terms.size() * sizeof(LeafTerm), hipMemcpyHostToDevice));
            }
        }
        ls.enabled = ok;
        ls.S = (int8_t)S;
    }
    // K-clique under a total order: every level's operand lists are the lists of
    // all earlier levels, each vertex is bounded below by its predecessor, and
    // nothing but adjacency filters the candidates.
    ctx->clique_k = 0;
    if (P >= 3 && ctx->info.simple && ctx->info.symmetric) {
        bool clique = true;
        for (int k = 1; k < P && clique; ++k) {
            const LevelPlan &lv = pv.lv[k];
            unsigned have = 1u << lv.src_level;
            for (int j = 0; j < lv.n_cons; ++j) have |= 1u << lv.cons_level[j];
            clique = have == (1u << k) - 1 && lv.n_upper == 0 && lv.label < 0;
            for (int t = 0; t < k; ++t) clique = clique && lt[t][k];
        }
        clique = clique && pv.lv[0].label < 0;
        if (clique) ctx->clique_k = P;
    }
    // The 4-cycle with its 8 automorphisms folded by ordering bounds, no labels:
    // four pattern edges, every vertex on two of them.
    ctx->cycle4 = false;
    if (P == 4 && ctx->info.simple && ctx->info.symmetric && p->n_aut == 8) {
        int deg[4] = {0, 0, 0, 0}, edges = 0;
        bool plain = pv.lv[0].label < 0;
        for (int k = 1; k < P; ++k) {
            const LevelPlan &lv = pv.lv[k];
            plain = plain && lv.label < 0;
            deg[lv.src_level]++, deg[k]++, edges++;
            for (int j = 0; j < lv.n_cons; ++j) deg[lv.cons_level[j]]++, deg[k]++, edges++;
        }
        ctx->cycle4 = plain && edges == 4 && deg[0] == 2 && deg[1] == 2 && deg[2] == 2 && deg[3] == 2;
    }
    // K6 minus a perfect matching (the octahedron) with its 48 automorphisms folded, no labels: twelve pattern
    // edges, every vertex on four of them (the only 4-regular graph on six vertices).
    ctx->octa = false;
    if (P == 6 && ctx->info.simple && ctx->info.symmetric && p->n_aut == 48) {
        int deg[6] = {0, 0, 0, 0, 0, 0}, edges = 0;
        bool plain = pv.lv[0].label < 0;
        for (int k = 1; k < P; ++k) {
            const LevelPlan &lv = pv.lv[k];
            plain = plain && lv.label < 0;
            deg[lv.src_level]++, deg[k]++, edges++;
            for (int j = 0; j < lv.n_cons; ++j) deg[lv.cons_level[j]]++, deg[k]++, edges++;
        }
        ctx->octa = plain && edges == 12;
        for (int k = 0; k < P; ++k) ctx->octa = ctx->octa && deg[k] == 4;
    }
    ctx->have_plan = true;
    return GPM_OK;
}

extern "C" int gpm_hip_set_cursor(gpm_ctx *ctx, volatile uint64_t *cursor) {
    if (!ctx) return fail(GPM_EINVAL, "null context");
    if ((ctx->shared_cursor == nullptr) != (cursor == nullptr)) ctx->chunks_valid = false;  // the chunk count depends on it
    ctx->shared_cursor = cursor;
    return GPM_OK;
}

extern "C" int gpm_hip_set_roots(gpm_ctx *ctx, int64_t lo, int64_t hi) {
    if (!ctx) return fail(GPM_EINVAL, "null context");
    if (!ctx->have_graph) return fail(GPM_ESTATE, "no graph uploaded");
    if (lo < 0 || hi < lo || hi > ctx->info.n) return fail(GPM_EINVAL, "root range out of bounds");
    ctx->root_lo = lo;
    ctx->root_hi = hi;
    ctx->chunks_valid = false;
    return GPM_OK;
}

extern "C" int gpm_hip_set_accounting(gpm_ctx *ctx, int on) {
    if (!ctx) return fail(GPM_EINVAL, "null context");
    ctx->accounting = on != 0;
    return GPM_OK;
}

extern "C" int gpm_hip_set_chunks(gpm_ctx *ctx, int64_t n_chunks) {
    if (!ctx) return fail(GPM_EINVAL, "null context");
    if (n_chunks < 0) return fail(GPM_EINVAL, "chunk count must not be negative");
    ctx->chunk_hint = n_chunks;
    ctx->chunks_valid = false;
    return GPM_OK;
}

// ---------------------------------------------------------------------------
// search driver
// ---------------------------------------------------------------------------

struct SearchArgs {
    gpm_ctx *ctx;
    bool count_only;
    bool direct_sink;  // single GPU: rows go straight to the sink on this thread
    gpm_row_sink sink;
    void *user;
    gpm_text_sink text_sink;  // non-null: matches leave as text lines formatted on the GPU
    bool gather = false;        // several workers, output mode: rows are gathered on GPU 0 in rounds
    bool fixed_memory = false;  // gpm_hip_count_noalloc: level-by-level and fused-tail kernels only (the clique and
                                // 4-cycle paths need large allocated buffers), roots cut into equal chunks
};

static int ensure_buffers(gpm_ctx *ctx, DevState &d, bool count_only) {
    HIP_TRY(hipSetDevice(d.dev));
    const int P = ctx->pv.P;
    int64_t want_rows = env_i64("GPM_ARENA_ROWS", (int64_t)1 << 24);
    int64_t min_rows = std::max<int64_t>((int64_t)std::max(ctx->info.max_out_deg, ctx->info.max_in_deg) + 64,
                                         ctx->chunk_roots);
    want_rows = std::max(want_rows, min_rows);
    int64_t want_emit = count_only ? 0 : std::max(env_i64("GPM_EMIT_ROWS", (int64_t)1 << 22), min_rows);
    if (d.cap_rows != want_rows) {
        free_buffers(d);
        d.cap_rows = want_rows;
        d.cap_emit = 0;
    }
    // frontiers F[k] hold rows of k+1 ids; levels 0..P-2 can be materialised
    for (int k = 0; k + 1 < std::max(P, 2); ++k) {
        if (!d.F[k]) HIP_TRY(hipMalloc(&d.F[k], (size_t)d.cap_rows * (size_t)(k + 1) * sizeof(int32_t)));
        if (!d.beg[k]) HIP_TRY(hipMalloc(&d.beg[k], (size_t)d.cap_rows * sizeof(int64_t)));
        if (!d.cnt[k]) HIP_TRY(hipMalloc(&d.cnt[k], (size_t)d.cap_rows * sizeof(int64_t)));
        if (!d.wsel[k]) HIP_TRY(hipMalloc(&d.wsel[k], (size_t)d.cap_rows));
        if (!d.pref[k]) {
            HIP_TRY(hipMalloc(&d.pref[k], ((size_t)d.cap_rows + 1) * sizeof(int64_t)));
            HIP_TRY(hipMemsetAsync(d.pref[k], 0, sizeof(int64_t), d.stream));
        }
    }
    if (ctx->t3.enabled && !d.over_rows) HIP_TRY(hipMalloc(&d.over_rows, 2 * (size_t)d.cap_rows * sizeof(int64_t)));
    if (!d.scan_tmp) {
        size_t bytes = 0;
        HIP_TRY(rocprim::inclusive_scan(nullptr, bytes, d.cnt[0], d.pref[0] + 1, (size_t)d.cap_rows,
                                        rocprim::plus<int64_t>(), d.stream));
        d.scan_tmp_bytes = bytes ? bytes : 8;
        HIP_TRY(hipMalloc(&d.scan_tmp, d.scan_tmp_bytes));
    }
    if (ctx->clique_k && !d.clq_slots) {
        d.clq_stride = ctx->chunk_roots;
        for (auto &sl : d.slots) {
            HIP_TRY(hipMalloc(&sl.lists, (size_t)CLQ_CLASSES * (size_t)d.clq_stride * sizeof(int32_t)));
            HIP_TRY(hipMalloc(&sl.rows, (size_t)CLQ_ROWS_WORDS * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&sl.counts, 16 * sizeof(unsigned long long)));
        }
        d.clq_scratch_blocks = (int)env_i64("GPM_CLIQUE_SCRATCH_BLOCKS", 512);
        for (auto &sl : d.slots)
            HIP_TRY(hipMalloc(&sl.scratch,
                              (size_t)d.clq_scratch_blocks * (size_t)CLQ_MAX_D * (CLQ_MAX_D / 32) * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&d.clq_slots, GPM_KERNEL_KINDS * CLQ_SLOTS * sizeof(unsigned long long)));
    }
    if (!count_only) {  // text output (allocated lazily by deliver_text) needs the automorphisms on the device
        dev_free(d.aut_dev);
        d.aut_dev = nullptr;
        const size_t bytes = std::max<size_t>(1, ctx->aut.size());
        HIP_TRY(hipMalloc(&d.aut_dev, bytes));
        if (!ctx->aut.empty())
            HIP_TRY(hipMemcpyAsync(d.aut_dev, ctx->aut.data(), ctx->aut.size(), hipMemcpyHostToDevice, d.stream));
    }
    const bool gather = !count_only && (ctx->devs.size() > 1 || !ctx->comms.empty()) && ctx->gather_on_device;
    if (gather && (want_emit > d.cap_emit || !d.ag_send)) {
        dev_free(d.ag_send);
        dev_free(d.ag_recv);
        dev_free(d.ag_counts);
        d.ag_send = d.ag_recv = nullptr;
        d.ag_counts = nullptr;
        const size_t row_bytes = (size_t)GPM_MAXP * sizeof(int32_t);  // room for any pattern set later
        HIP_TRY(hipMalloc(&d.ag_send, (size_t)want_emit * row_bytes));
        // with communicators every GPU receives everybody's rows (all-gather); without, only GPU 0 collects
        if (d.index == 0 || !ctx->comms.empty())
            HIP_TRY(hipMalloc(&d.ag_recv, ctx->devs.size() * (size_t)want_emit * row_bytes));
        HIP_TRY(hipMalloc(&d.ag_counts, (1 + ctx->devs.size()) * sizeof(unsigned long long)));
    }
    if (want_emit > d.cap_emit) {
        dev_free(d.emit);
        host_free(d.emit_host);
        d.emit = nullptr;
        d.emit_host = nullptr;
        HIP_TRY(hipMalloc(&d.emit, (size_t)want_emit * (size_t)P * sizeof(int32_t)));
        HIP_TRY(hipHostMalloc(&d.emit_host, (size_t)want_emit * (size_t)P * sizeof(int32_t)));
        d.cap_emit = want_emit;
    }
    return GPM_OK;
}

static int read_ctr(DevState &d, int slot, uint64_t *out) {
    HIP_TRY(hipMemcpyAsync(d.ctr_host + slot, d.ctr + slot, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           d.stream));
    HIP_TRY(hipStreamSynchronize(d.stream));
    *out = d.ctr_host[slot];
    return GPM_OK;
}

static int timed_begin(DevState &d, int kind, int64_t rows = 0, hipStream_t stream = nullptr, int tag = 0) {
    if (d.fixed) return GPM_OK;
    if (d.events_used == d.events.size()) {
        EventPair ev{};
        ev.kind = kind;
        HIP_TRY(hipEventCreate(&ev.a));
        HIP_TRY(hipEventCreate(&ev.b));
        d.events.push_back(ev);
    }
    d.events[d.events_used].kind = kind;
    d.events[d.events_used].rows = rows;
    d.events[d.events_used].tag = tag;
    HIP_TRY(hipEventRecord(d.events[d.events_used].a, stream ? stream : d.stream));
    return GPM_OK;
}

static int timed_end(DevState &d, hipStream_t stream = nullptr) {
    if (d.fixed) return GPM_OK;
    HIP_TRY(hipEventRecord(d.events[d.events_used].b, stream ? stream : d.stream));
    d.events_used++;
    return GPM_OK;
}

// Turn timing events into milliseconds; the stream must be idle.
static int collect_times(DevState &d, bool trace) {
    for (size_t i = 0; i < d.events_used; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, d.events[i].a, d.events[i].b));
        if (trace)
            fprintf(stderr, "gpm trace: %s class %d rows %lld ms %.3f\n", kKindNames[d.events[i].kind],
                    d.events[i].tag, (long long)d.events[i].rows, ms);
        d.ms_kind[d.events[i].kind] += ms;
        d.launches_kind[d.events[i].kind]++;
    }
    d.events_used = 0;
    return GPM_OK;
}

// Text mode: the canonical rows emit[0, nrows) on the device become text lines
// there (every automorphic image one line) and reach the sink in blocks of
// whole lines.  Blocks are double-buffered: while the sink consumes one, the
// next is formatted and copied.
static int deliver_text(const SearchArgs &a, DevState &d, const int32_t *rows_dev, uint64_t nrows) {
    gpm_ctx *ctx = a.ctx;
    const int P = ctx->pv.P;
    const int64_t n_aut = (int64_t)ctx->plan.n_aut;
    hipStream_t s = d.stream;
    if (nrows == 0) return GPM_OK;
    const int64_t line_max = 11 * (int64_t)P;
    if (!d.text_dev || d.text_line_max != line_max) {
        dev_free(d.line_off);
        dev_free(d.text_dev);
        dev_free(d.text_scan_tmp);
        host_free(d.text_host[0]);
        host_free(d.text_host[1]);
        d.line_off = nullptr, d.text_dev = nullptr, d.text_host[0] = d.text_host[1] = nullptr;
        d.text_scan_tmp = nullptr;
        d.text_lines_cap = std::max<int64_t>(n_aut, env_i64("GPM_TEXT_LINES", (int64_t)1 << 23));
        d.text_line_max = line_max;
        const size_t bytes = (size_t)d.text_lines_cap * (size_t)line_max;
        HIP_TRY(hipMalloc(&d.line_off, ((size_t)d.text_lines_cap + 1) * sizeof(int64_t)));
        HIP_TRY(hipMalloc(&d.text_dev, bytes));
        HIP_TRY(hipHostMalloc(&d.text_host[0], bytes));
        HIP_TRY(hipHostMalloc(&d.text_host[1], bytes));
        d.text_scan_bytes = 0;
        HIP_TRY(rocprim::inclusive_scan(nullptr, d.text_scan_bytes, d.line_off + 1, d.line_off + 1,
                                        (size_t)d.text_lines_cap, rocprim::plus<int64_t>(), s));
        if (!d.text_scan_bytes) d.text_scan_bytes = 8;
        HIP_TRY(hipMalloc(&d.text_scan_tmp, d.text_scan_bytes));
    }
    const int64_t rows_per_block = std::max<int64_t>(1, d.text_lines_cap / n_aut);
    int cur = 0;
    size_t pending_bytes = 0;   // block sitting in text_host[cur ^ 1], copy complete, not yet given to the sink
    uint64_t pending_lines = 0;
    auto flush = [&](int buf, size_t bytes, uint64_t lines) -> int {
        if (!bytes) return GPM_OK;
        if (a.direct_sink) {
            if (a.text_sink(a.user, d.text_host[buf], bytes, lines) != 0) {
                ctx->stop = true;
                return fail(GPM_ESINK, "the sink stopped the search");
            }
            return GPM_OK;
        }
        RowBatch batch;
        batch.nrows = lines;
        batch.text.assign(d.text_host[buf], d.text_host[buf] + bytes);
        std::unique_lock<std::mutex> lk(ctx->mu);
        ctx->cv.wait(lk, [&] { return ctx->queue.size() < 8 || ctx->stop; });
        if (ctx->stop) return fail(GPM_ESINK, "the sink stopped the search");
        ctx->queue.push_back(std::move(batch));
        ctx->cv.notify_all();
        return GPM_OK;
    };
    for (uint64_t r0 = 0; r0 < nrows && !ctx->stop; r0 += (uint64_t)rows_per_block) {
        const int64_t nr = (int64_t)std::min<uint64_t>((uint64_t)rows_per_block, nrows - r0);
        const int64_t nl = nr * n_aut;
        const int32_t *rows = rows_dev + r0 * (uint64_t)P;
        k_line_len<<<grid_for(nl), 256, 0, s>>>(rows, nr, P, (int)n_aut, d.aut_dev, d.line_off + 1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetAsync(d.line_off, 0, sizeof(int64_t), s));
        size_t tmp_bytes = d.text_scan_bytes;
        HIP_TRY(rocprim::inclusive_scan(d.text_scan_tmp, tmp_bytes, d.line_off + 1, d.line_off + 1, (size_t)nl,
                                        rocprim::plus<int64_t>(), s));
        HIP_TRY(hipMemcpyAsync(d.i64_host, d.line_off + nl, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const size_t bytes = (size_t)d.i64_host[0];
        k_line_write<<<grid_for(nl), 256, 0, s>>>(rows, nr, P, (int)n_aut, d.aut_dev, d.line_off, d.text_dev);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(d.text_host[cur], d.text_dev, bytes, hipMemcpyDeviceToHost, s));
        int rc = flush(cur ^ 1, pending_bytes, pending_lines);  // the sink works while the copy runs
        if (rc) {
            (void)hipStreamSynchronize(s);
            return rc;
        }
        HIP_TRY(hipStreamSynchronize(s));
        pending_bytes = bytes;
        pending_lines = (uint64_t)nl;
        cur ^= 1;
    }
    return flush(cur ^ 1, pending_bytes, pending_lines);
}

// Hand finished rows (pattern-vertex order, 1-based ids) to the sink, after
// expanding each into its automorphic images.
static int deliver(const SearchArgs &a, DevState &d, const int32_t *rows, uint64_t nrows) {
    gpm_ctx *ctx = a.ctx;
    const int P = ctx->pv.P;
    const uint64_t n_aut = ctx->plan.n_aut;
    const int8_t *aut = ctx->plan.aut;
    (void)d;
    if (nrows == 0) return GPM_OK;
    const uint64_t step = std::max<uint64_t>(1, ((uint64_t)1 << 20) / n_aut);
    for (uint64_t r0 = 0; r0 < nrows; r0 += step) {
        uint64_t r1 = std::min(nrows, r0 + step);
        RowBatch batch;
        batch.nrows = (r1 - r0) * n_aut;
        batch.rows.resize((size_t)batch.nrows * (size_t)P);
        int32_t *o = batch.rows.data();
        for (uint64_t r = r0; r < r1; ++r) {
            const int32_t *m = rows + r * (uint64_t)P;
            if (n_aut == 1) {
                memcpy(o, m, (size_t)P * sizeof(int32_t));
                o += P;
                continue;
            }
            for (uint64_t i = 0; i < n_aut; ++i, o += P)
                for (int v = 0; v < P; ++v) o[v] = m[aut[i * (uint64_t)P + (uint64_t)v]];
        }
        if (a.direct_sink) {
            if (a.sink && a.sink(a.user, batch.rows.data(), batch.nrows, P) != 0) {
                ctx->stop = true;
                return fail(GPM_ESINK, "the row sink stopped the search");
            }
        } else {
            std::unique_lock<std::mutex> lk(ctx->mu);
            ctx->cv.wait(lk, [&] { return ctx->queue.size() < 8 || ctx->stop; });
            if (ctx->stop) return fail(GPM_ESINK, "the row sink stopped the search");
            ctx->queue.push_back(std::move(batch));
            ctx->cv.notify_all();
        }
    }
    return GPM_OK;
}

// ---- match lists of several GPUs, gathered on GPU 0 -------------------------
//
// The reference sends every match to rank 0 as one message and rank 0 writes it
// (reference: src/gpm-par.c:181, :608-621).  Here every GPU of a context keeps
// its finished rows in a send buffer; from time to time all workers meet in a
// gather round: the row counts are exchanged, the lists padded to the longest
// one are all-gathered over RCCL (every GPU ends up with everybody's rows), and
// the worker of GPU 0 hands its copy to the writer -- as rows through a pinned
// buffer, or formatted into text lines on GPU 0.  A context without
// communicators (GPM_NO_RCCL=1) collects the lists with peer copies into GPU 0
// instead.  A round forms when a worker's send buffer cannot take its next
// batch (it asks for one and every other worker joins at its next check point)
// and once more when every worker has run out of chunks.

// Meet the other workers; returns through `rows` what each brings and whether
// this is the round after which everybody is done.
static int gather_meet(gpm_ctx *ctx, DevState &d, uint64_t rows[8], bool *last) {
    auto &gs = ctx->gs;
    const int n = (int)ctx->devs.size();
    std::unique_lock<std::mutex> lk(ctx->mu);
    const uint64_t gen = gs.generation;
    const int par = (int)(gen & 1);
    gs.rows[par][d.index] = d.ag_rows;
    if (++gs.arrived == n) {
        gs.last[par] = gs.finished == n;
        gs.arrived = 0;
        gs.requested = false;
        gs.generation++;
        gs.rounds++;
        ctx->cv.notify_all();
    } else {
        ctx->cv.wait(lk, [&] { return gs.generation != gen || ctx->stop; });
        if (gs.generation == gen) return fail(GPM_ESINK, "the search was stopped");
    }
    for (int i = 0; i < n; ++i) rows[i] = gs.rows[par][i];
    *last = gs.last[par];
    return GPM_OK;
}

// Plain barrier of the workers inside a round (its own generation counter: the round's bookkeeping --
// rounds, rows, requested -- is not touched).
static int gather_wait(gpm_ctx *ctx) {
    auto &gs = ctx->gs;
    const int n = (int)ctx->devs.size();
    std::unique_lock<std::mutex> lk(ctx->mu);
    const uint64_t gen = gs.wait_generation;
    if (++gs.waiting == n) {
        gs.waiting = 0;
        gs.wait_generation++;
        ctx->cv.notify_all();
        return GPM_OK;
    }
    ctx->cv.wait(lk, [&] { return gs.wait_generation != gen || ctx->stop; });
    if (gs.wait_generation == gen) return fail(GPM_ESINK, "the search was stopped");
    return GPM_OK;
}

static int deliver(const SearchArgs &a, DevState &d, const int32_t *rows, uint64_t nrows);
static int deliver_text(const SearchArgs &a, DevState &d, const int32_t *rows_dev, uint64_t nrows);

// One gather round, run by every worker.  *last tells the caller that no further round follows.
static int gather_round(const SearchArgs &a, DevState &d, bool *last) {
    gpm_ctx *ctx = a.ctx;
    const int n = (int)ctx->devs.size();
    const int P = ctx->pv.P;
    hipStream_t s = d.stream;
    HIP_TRY(hipStreamSynchronize(s));  // this worker's rows are in its send buffer
    uint64_t rows[8];
    int rc = gather_meet(ctx, d, rows, last);
    if (rc) return rc;
    uint64_t longest = 0, all = 0;
    for (int i = 0; i < n; ++i) {
        longest = std::max(longest, rows[i]);
        all += rows[i];
    }
    DevState &d0 = ctx->devs[0];
    if (all) {
        // What a worker has to do before the exchange (here: its row count on the device) can fail, and a
        // collective that one rank never joins blocks the others for good.  So every worker reports how its
        // preparation went, all meet once more, and the exchange is issued only if nobody failed.
        // GPM_FAULT_GATHER=<worker> makes that worker report a failure in its first round (tests).
        bool prep_ok = true;
        if (!ctx->comms.empty()) {
            d.ctr_host[CTR_CLQ_CLASS0] = d.ag_rows;
            prep_ok = hipMemcpyAsync(d.ag_counts, d.ctr_host + CTR_CLQ_CLASS0, sizeof(unsigned long long),
                                     hipMemcpyHostToDevice, s) == hipSuccess;
        }
        const int64_t fault_worker = env_i64("GPM_FAULT_GATHER", -1);
        if (fault_worker == d.index) prep_ok = false;
        if (!prep_ok) {
            std::lock_guard<std::mutex> lk(ctx->mu);
            ctx->gs.round_failed = true;
        }
        if ((rc = gather_wait(ctx))) return rc;
        if (ctx->gs.round_failed) {
            ctx->stop = true;
            ctx->cv.notify_all();
            return fail(GPM_ERUNTIME, "worker %d: a worker could not prepare its part of a gather round", d.index);
        }
        if (!ctx->comms.empty()) {
            const Rccl &R = rccl();
            // row counts first (every GPU learns how much of each padded list is real), then the lists
            int nrc = R.AllGather(d.ag_counts, d.ag_counts + 1, 1, RCCL_UINT64, ctx->comms[(size_t)d.index], s);
            if (nrc == 0)
                nrc = R.AllGather(d.ag_send, d.ag_recv, (size_t)longest * (size_t)P, RCCL_INT32,
                                  ctx->comms[(size_t)d.index], s);
            if (nrc != 0) return fail(GPM_ERUNTIME, "RCCL all-gather of the match lists failed: %s", R.GetErrorString(nrc));
            if (d.index == 0) {
                unsigned long long got[8] = {};
                HIP_TRY(hipMemcpyAsync(got, d.ag_counts + 1, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                for (int i = 0; i < n; ++i)
                    if (got[i] != rows[i])
                        return fail(GPM_ERUNTIME, "gathered row count of GPU %d is %llu, its worker brought %llu", i, got[i],
                                    (unsigned long long)rows[i]);
            } else {
                HIP_TRY(hipStreamSynchronize(s));
            }
        } else if (d.index == 0) {
            for (int i = 0; i < n; ++i) {
                if (!rows[i]) continue;
                DevState &src = ctx->devs[(size_t)i];
                HIP_TRY(hipMemcpyPeerAsync(d0.ag_recv + (size_t)i * (size_t)longest * (size_t)P, d0.dev, src.ag_send, src.dev,
                                           (size_t)rows[i] * (size_t)P * sizeof(int32_t), s));
            }
            HIP_TRY(hipStreamSynchronize(s));
        }
    }
    d.ag_rows = 0;
    // peer copies read the others' send buffers: nobody refills before GPU 0 is done
    if (ctx->comms.empty() && (rc = gather_wait(ctx))) return rc;
    if (d.index == 0 && all) {
        ctx->gs.gathered_rows += all;
        for (int i = 0; i < n && !ctx->stop; ++i) {
            const int32_t *part = d0.ag_recv + (size_t)i * (size_t)longest * (size_t)P;
            if (!rows[i]) continue;
            if (a.text_sink) {
                if ((rc = deliver_text(a, d0, part, rows[i]))) return rc;
                continue;
            }
            for (uint64_t r0 = 0; r0 < rows[i]; r0 += (uint64_t)d0.cap_emit) {
                const uint64_t nr = std::min<uint64_t>((uint64_t)d0.cap_emit, rows[i] - r0);
                HIP_TRY(hipMemcpyAsync(d0.emit_host, part + r0 * (uint64_t)P, (size_t)nr * (size_t)P * sizeof(int32_t),
                                       hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                if ((rc = deliver(a, d0, d0.emit_host, nr))) return rc;
            }
        }
    }
    return GPM_OK;
}

// Check point of a worker in gather mode: join the round somebody asked for.
static int gather_check(const SearchArgs &a, DevState &d) {
    if (!a.gather || !a.ctx->gs.requested.load()) return GPM_OK;
    bool last;
    return gather_round(a, d, &last);
}

// `n_emit` finished rows lie in d.emit on the device: pass them on -- to the send buffer of the gather
// rounds, to the text formatter, or through the pinned buffer to the row sink.
static int hand_over(const SearchArgs &a, DevState &d, uint64_t n_emit) {
    gpm_ctx *ctx = a.ctx;
    const int P = ctx->pv.P;
    hipStream_t s = d.stream;
    if (!n_emit) return GPM_OK;
    if (a.gather) {
        while (d.ag_rows + n_emit > (uint64_t)d.cap_emit) {  // no room: a round empties every send buffer
            {
                std::lock_guard<std::mutex> lk(ctx->mu);  // workers that have run out of chunks sleep on the flag
                ctx->gs.requested = true;
                ctx->cv.notify_all();
            }
            bool last;
            int rc = gather_round(a, d, &last);
            if (rc) return rc;
        }
        HIP_TRY(hipMemcpyAsync(d.ag_send + d.ag_rows * (uint64_t)P, d.emit, (size_t)n_emit * (size_t)P * sizeof(int32_t),
                               hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));  // d.emit is free for the next batch
        d.ag_rows += n_emit;
        return gather_check(a, d);
    }
    if (a.text_sink) return deliver_text(a, d, d.emit, n_emit);
    HIP_TRY(hipMemcpyAsync(d.emit_host, d.emit, (size_t)n_emit * (size_t)P * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return deliver(a, d, d.emit_host, n_emit);
}

// A worker that has run out of chunks keeps joining rounds until the one in which everybody is done.
static int gather_finish(const SearchArgs &a, DevState &d) {
    gpm_ctx *ctx = a.ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->gs.finished++;
        if (ctx->gs.finished == (int)ctx->devs.size()) ctx->gs.requested = true;  // the closing round
        ctx->cv.notify_all();
    }
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(ctx->mu);
            ctx->cv.wait(lk, [&] { return ctx->gs.requested.load() || ctx->stop; });
            if (ctx->stop && !ctx->gs.requested.load()) return fail(GPM_ESINK, "the search was stopped");
        }
        bool last = false;
        int rc = gather_round(a, d, &last);
        if (rc) return rc;
        if (last) return GPM_OK;
    }
}

// Expand the partial matches F[k-1][0..nrows) (k ids each) to completion.
static int expand(const SearchArgs &a, DevState &d, int k, int64_t nrows) {
    gpm_ctx *ctx = a.ctx;
    const int P = ctx->pv.P;
    hipStream_t s = d.stream;
    if (nrows == 0 || ctx->stop) return GPM_OK;
    d.rows_in[k - 1] += (uint64_t)nrows;
    const int32_t *rows = d.F[k - 1];
    progress_tick(d.dev, d.chunks, ctx->chunks.size(), k, nrows);

    if (a.count_only && ctx->use_tail && ctx->ls.enabled && k == ctx->ls.S) {
        const unsigned blocks = (unsigned)std::min<int64_t>((nrows + TAIL_WAVES - 1) / TAIL_WAVES, 16384);
        int rc = timed_begin(d, GPM_KIND_TAIL, nrows);
        if (rc) return rc;
        k_leaf_suffix<<<blocks, TAIL_WAVES * GPM_WAVE, 0, s>>>(d.gv, ctx->pv, ctx->ls, d.ls_terms, rows, nrows,
                                                              d.ctr + CTR_TAIL_MATCHES, d.ctr + CTR_TAIL_BYTES);
        HIP_TRY(hipGetLastError());
        return timed_end(d);
    }
    if (a.count_only && ctx->use_tail && k == P - 3 && ctx->t3.enabled && !d.no_fuse3) {
        if (!d.over_rows) return fail(GPM_ESTATE, "hand-back lists of the three-level kernel were not allocated");
        int64_t *over1 = d.over_rows, *over2 = d.over_rows + d.cap_rows;
        HIP_TRY(hipMemsetAsync(d.ctr + CTR_OVER, 0, 3 * sizeof(unsigned long long), s));
        // 8 lanes and sets of up to 64 per partial match; what does not fit goes to 16 lanes with sets of up
        // to 256, then to whole waves with sets of up to 1024; what still does not fit goes level by level
        const int64_t grp = ctx->fuse3_group;
        unsigned blocks = (unsigned)std::min<int64_t>((nrows * grp + 255) / 256, 1 << 16);
        int rc = timed_begin(d, GPM_KIND_TAIL, nrows);
        if (rc) return rc;
        if (grp == 4)
            k_tail3<4, 64><<<blocks, 256, 0, s>>>(d.gv, ctx->pv, ctx->ti, ctx->t3, rows, nullptr, nrows, nullptr, over1,
                                                 d.ctr + CTR_OVER, d.ctr + CTR_TAIL_MATCHES, d.ctr + CTR_TAIL_BYTES);
        else if (grp == 8)
            k_tail3<8, 64><<<blocks, 256, 0, s>>>(d.gv, ctx->pv, ctx->ti, ctx->t3, rows, nullptr, nrows, nullptr, over1,
                                                 d.ctr + CTR_OVER, d.ctr + CTR_TAIL_MATCHES, d.ctr + CTR_TAIL_BYTES);
        else if (grp == 32)
            k_tail3<32, 64><<<blocks, 256, 0, s>>>(d.gv, ctx->pv, ctx->ti, ctx->t3, rows, nullptr, nrows, nullptr, over1,
                                                  d.ctr + CTR_OVER, d.ctr + CTR_TAIL_MATCHES, d.ctr + CTR_TAIL_BYTES);
        else
            k_tail3<16, 64><<<blocks, 256, 0, s>>>(d.gv, ctx->pv, ctx->ti, ctx->t3, rows, nullptr, nrows, nullptr, over1,
                                                  d.ctr + CTR_OVER, d.ctr + CTR_TAIL_MATCHES, d.ctr + CTR_TAIL_BYTES);
        HIP_TRY(hipGetLastError());
        if ((rc = timed_end(d))) return rc;
        uint64_t n_over = 0;
        if (!ctx->fuse3_wave) {  // tests: straight to the level-by-level path
            if ((rc = read_ctr(d, CTR_OVER, &n_over))) return rc;
            if (n_over == 0) return GPM_OK;
            over2 = over1;
        } else {
            // The wider tiers are enqueued behind the first without the host looking at the hand-back counts:
            // each reads the number of its partial matches from the counter the tier before it left (a tier
            // with nothing to do costs one empty launch), and the host reads one count, the last.  The grids
            // are sized for the worst case the kernels' grid-stride loops make harmless: 2048 workgroups, more than fit the GPU at once.
            const unsigned tier_blocks = 2048;
            const unsigned long long *from = d.ctr + CTR_OVER;
            if (ctx->fuse3_mid) {  // sets of up to 256 with 16 lanes
                if ((rc = timed_begin(d, GPM_KIND_TAIL, 0))) return rc;
                k_tail3<16, 256><<<tier_blocks, 256, 0, s>>>(d.gv, ctx->pv, ctx->ti, ctx->t3, rows, over1, 0, from, over2,
                                                            d.ctr + CTR_OVER + 1, d.ctr + CTR_TAIL_MATCHES,
                                                            d.ctr + CTR_TAIL_BYTES);
                HIP_TRY(hipGetLastError());
                if ((rc = timed_end(d))) return rc;
                std::swap(over1, over2);  // over1: what is left; over2 free again
                from = d.ctr + CTR_OVER + 1;
            }
            // whole waves, sets of up to 1024; its own hand-back count goes to a third counter
            if ((rc = timed_begin(d, GPM_KIND_TAIL, 0))) return rc;
            k_tail3<64, 1024><<<tier_blocks, 256, 0, s>>>(d.gv, ctx->pv, ctx->ti, ctx->t3, rows, over1, 0, from, over2,
                                                         d.ctr + CTR_OVER + 2, d.ctr + CTR_TAIL_MATCHES,
                                                         d.ctr + CTR_TAIL_BYTES);
            HIP_TRY(hipGetLastError());
            if ((rc = timed_end(d))) return rc;
            if ((rc = read_ctr(d, CTR_OVER + 2, &n_over))) return rc;
            if (n_over == 0) return GPM_OK;
        }
        // the partial matches whose sets did not fit a lane group: compact them to the front of this
        // frontier (through the next level's arena) and expand them level by level
        k_gather_rows<<<grid_for((int64_t)n_over * k), 256, 0, s>>>(rows, over2, (int64_t)n_over, k, d.F[k]);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(d.F[k - 1], d.F[k], (size_t)n_over * (size_t)k * sizeof(int32_t),
                               hipMemcpyDeviceToDevice, s));
        d.rows_in[k - 1] -= n_over;  // counted again by the call below
        d.no_fuse3 = true;
        rc = expand(a, d, k, (int64_t)n_over);
        d.no_fuse3 = false;
        return rc;
    }
    if (a.count_only && ctx->use_tail && k == P - 2 && ctx->ti.leaf_pair) {
        int rc = timed_begin(d, GPM_KIND_TAIL, nrows);
        if (rc) return rc;
        k_leaf_pair<<<grid_for(nrows), 256, 0, s>>>(d.gv, ctx->pv, ctx->ti, rows, nrows, d.ctr + CTR_TAIL_MATCHES,
                                                    d.ctr + CTR_TAIL_BYTES);
        HIP_TRY(hipGetLastError());
        return timed_end(d);
    }
    if (a.count_only && ctx->use_tail && k == P - 2 && ctx->ti.twin && ctx->twin_group > 0) {
        const int64_t G = ctx->twin_group;
        const unsigned blocks = (unsigned)std::min<int64_t>((nrows * G + 255) / 256, 1 << 16);
        int rc = timed_begin(d, GPM_KIND_TAIL, nrows);
        if (rc) return rc;
#define TWIN_LAUNCH(GG)                                                                                        \
    k_tail_twin<GG><<<blocks, 256, 0, s>>>(d.gv, ctx->pv, ctx->ti, rows, nrows, d.ctr + CTR_TAIL_MATCHES, \
                                           d.ctr + CTR_TAIL_BYTES)
        if (G >= 64) TWIN_LAUNCH(64);
        else if (G >= 16) TWIN_LAUNCH(16);
        else if (G >= 8) TWIN_LAUNCH(8);
        else if (G >= 4) TWIN_LAUNCH(4);
        else if (G >= 2) TWIN_LAUNCH(2);
        else TWIN_LAUNCH(1);
#undef TWIN_LAUNCH
        HIP_TRY(hipGetLastError());
        return timed_end(d);
    }
    if (a.count_only && ctx->use_tail && k == P - 2) {
        unsigned blocks = (unsigned)std::min<int64_t>((nrows + TAIL_WAVES - 1) / TAIL_WAVES, 16384);
        int rc = timed_begin(d, GPM_KIND_TAIL, nrows);
        if (rc) return rc;
        k_tail<<<blocks, TAIL_WAVES * GPM_WAVE, 0, s>>>(d.gv, ctx->pv, ctx->ti, rows, nrows, d.ctr + CTR_TAIL_MATCHES,
                                                        d.ctr + CTR_TAIL_BYTES);
        HIP_TRY(hipGetLastError());
        d.rows_in[k] += 0;  // rows of the fused level never materialise
        return timed_end(d);
    }

    // candidate ranges and their prefix sums
    k_ranges<<<grid_for(nrows), 256, 0, s>>>(d.gv, ctx->pv, k, rows, nrows, d.beg[k - 1], d.cnt[k - 1], d.wsel[k - 1],
                                             d.ctr + CTR_BYTES);
    HIP_TRY(hipGetLastError());
    size_t tmp_bytes = d.scan_tmp_bytes;
    HIP_TRY(rocprim::inclusive_scan(d.scan_tmp, tmp_bytes, d.cnt[k - 1], d.pref[k - 1] + 1, (size_t)nrows,
                                    rocprim::plus<int64_t>(), s));
    HIP_TRY(hipMemcpyAsync(d.i64_host, d.pref[k - 1] + nrows, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int64_t total = d.i64_host[0];
    if (total == 0) return GPM_OK;

    const bool terminal = k == P - 1;
    const int64_t cap = terminal ? (a.count_only ? ((int64_t)1 << 30) : d.cap_emit) : d.cap_rows;
    int64_t b0 = 0, g0 = 0;
    while (b0 < nrows && !ctx->stop) {
        int64_t b1 = nrows, gcount = total - g0;
        if (gcount > cap) {
            k_batch_end<<<1, 1, 0, s>>>(d.pref[k - 1], nrows, b0, cap, d.i64_dev);
            HIP_TRY(hipMemcpyAsync(d.i64_host, d.i64_dev, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            b1 = d.i64_host[0];
            gcount = d.i64_host[1];
            if (gcount > cap) return fail(GPM_ERUNTIME, "one candidate list exceeds the frontier arena");
        }
        if (gcount > 0) {
            const int64_t g1 = g0 + gcount;
            int rc = timed_begin(d, GPM_KIND_EXTEND);
            if (rc) return rc;
            if (terminal && a.count_only) {
                k_extend<MODE_COUNT><<<grid_for(gcount), 256, 0, s>>>(d.gv, ctx->pv, k, rows, d.pref[k - 1],
                                                                      d.beg[k - 1], d.wsel[k - 1], b0, b1, g0, g1, nullptr,
                                                                      d.ctr + CTR_MATCHES, nullptr);
                HIP_TRY(hipGetLastError());
                rc = timed_end(d);
                if (rc) return rc;
            } else if (terminal) {
                HIP_TRY(hipMemsetAsync(d.ctr + CTR_EMIT, 0, sizeof(unsigned long long), s));
                k_extend<MODE_EMIT><<<grid_for(gcount), 256, 0, s>>>(d.gv, ctx->pv, k, rows, d.pref[k - 1],
                                                                     d.beg[k - 1], d.wsel[k - 1], b0, b1, g0, g1, d.emit,
                                                                     d.ctr + CTR_EMIT, d.perm);
                HIP_TRY(hipGetLastError());
                rc = timed_end(d);
                if (rc) return rc;
                uint64_t n_emit = 0;
                rc = read_ctr(d, CTR_EMIT, &n_emit);
                if (rc) return rc;
                d.canonical += n_emit;
                if ((rc = hand_over(a, d, n_emit))) return rc;
            } else {
                HIP_TRY(hipMemsetAsync(d.ctr + CTR_LEVEL0 + k, 0, sizeof(unsigned long long), s));
                k_extend<MODE_EXTEND><<<grid_for(gcount), 256, 0, s>>>(d.gv, ctx->pv, k, rows, d.pref[k - 1],
                                                                       d.beg[k - 1], d.wsel[k - 1], b0, b1, g0, g1, d.F[k],
                                                                       d.ctr + CTR_LEVEL0 + k, nullptr);
                HIP_TRY(hipGetLastError());
                rc = timed_end(d);
                if (rc) return rc;
                uint64_t n_next = 0;
                rc = read_ctr(d, CTR_LEVEL0 + k, &n_next);
                if (rc) return rc;
                rc = expand(a, d, k + 1, (int64_t)n_next);
                if (rc) return rc;
            }
            g0 = g1;
        }
        b0 = b1;
    }
    return GPM_OK;
}

// ---- clique fast path ------------------------------------------------------

// Dynamic LDS of k_clique<K, BLOCK, rows_in_lds>; must mirror the kernel's layout.
static size_t clq_lds_bytes(int K, int block, int dcap, bool rows_in_lds) {
    const size_t wcap = dcap / 32 < 4 ? 4 : dcap / 32;
    const size_t depths = K > 3 ? K - 3 : 1;
    const size_t work = std::max((size_t)(block / GPM_WAVE) * ((size_t)dcap / 2 + depths * wcap), 2 * (size_t)dcap);
    size_t words = 2 * (size_t)dcap + work + (rows_in_lds ? (size_t)dcap * wcap : 0);
    return words * sizeof(uint32_t);
}

// Classes 2..5 of a chunk through the build/count kernel pair (K = 3: the
// build kernel counts the triangles itself; K = 4: tiled count).  all_hub: every
// root of the chunk (and so every vertex of its universes) lies in the hub
// matrix; stats: sum the merge-model bytes as well.
template <int K, bool STATS, bool ALL_HUB>
static int clique_launch_big(DevState &d, ChunkSlot &sl, const uint32_t n[CLQ_CLASSES], int bias, int64_t chunk_hi) {
    static_assert(K == 3 || K == 4, "build/count pair exists for triangles and 4-cliques");
    hipStream_t s = sl.stream;
    unsigned long long *cnt = d.clq_slots, *byt = d.clq_slots + (size_t)GPM_KIND_CLQ_BUILD * CLQ_SLOTS;
    unsigned long long *byt3 = d.clq_slots + (size_t)GPM_KIND_CLQ_COUNT * CLQ_SLOTS;  // level 3, done by the count kernel
    // universes of this chunk lie above its lowest root, vertex n - chunk_hi: the id range an LDS bitmap
    // of a universe has to cover (k_clq_build_lists), in words rounded up to a multiple of 1024
    const int32_t base_id = (int32_t)std::max<int64_t>(0, (int64_t)d.n - chunk_hi);
    const int bm_words = (int)((((int64_t)d.n - base_id + 31) / 32 + 1023) / 1024 * 1024);
    const int dcaps[4] = {512, 1024, 2048, 4096};
    const int64_t strides[4] = {512 * 16, 1024 * 32, (1024 + 2048) * 32, (1024 + 2048 + 3072 + 4096) * 32};
    int rc;
    for (int c = 5; c >= 2; --c) {  // longest-running class first
        if (!n[c]) continue;
        const int dcap = dcaps[c - 2];
        const int64_t stride = strides[c - 2];
        const int64_t batch = (int64_t)(CLQ_ROWS_WORDS / (size_t)stride);
        const int32_t *list = sl.lists + (int64_t)c * d.clq_stride;
        const size_t lds_build = ((size_t)(ALL_HUB && !STATS ? 1 : 4) * dcap + 4 * 128) * sizeof(uint32_t);
        // 16 waves hide the LDS latency of the big tiles better than 8; the J lists of 16 waves and a
        // 128 KiB tile fill the 160 KiB of LDS exactly
        const int vt = std::min(dcap, d.count4_vt[c - 2]);  // columns per tile of the count
        const bool wide = vt >= d.count4_wide_min && d.count4_wide;
        const int nw = wide ? 16 : 8, seg = wide ? 2 : 4;
        const size_t lds_count = ((size_t)nw * (vt / 2) + (size_t)vt * (vt / 32)) * sizeof(uint32_t);
        const int tcap = dcap / vt, phases = tcap * (tcap + 1) / 2;
        for (int64_t b0 = 0; b0 < (int64_t)n[c]; b0 += batch) {
            const int64_t nb = std::min<int64_t>(batch, (int64_t)n[c] - b0);
            if ((rc = timed_begin(d, GPM_KIND_CLQ_BUILD, nb, s, c))) return rc;
            if (ALL_HUB && !STATS && c <= d.clq_lists_max_class && bm_words <= CLQ_BM_MAX_WORDS) {
                // rows from the adjacency lists, tested against a bitmap of the universe in LDS
                const int RB = d.clq_build_rb > 0 ? d.clq_build_rb : (c == 2 ? 1 : c == 3 ? 2 : c == 4 ? 4 : 8);
                const int nwl = d.clq_lists_waves;
                const size_t lds_lists = ((size_t)bm_words + bm_words / 4 + 2 * (size_t)dcap + 256 + (size_t)nwl * 128) * sizeof(uint32_t);
                if (nwl == 16) {
                    HIP_TRY(hipFuncSetAttribute((const void *)k_clq_build_lists<K, 16>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lists));
                    k_clq_build_lists<K, 16><<<(unsigned)(nb * RB), 1024, lds_lists, s>>>(
                        d.gv, list + b0, nb, dcap, RB, base_id, bm_words, d.clq_kappa4[c - 2], d.clq_probe, sl.rows, stride, cnt);
                } else {
                    HIP_TRY(hipFuncSetAttribute((const void *)k_clq_build_lists<K, 4>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lists));
                    k_clq_build_lists<K, 4><<<(unsigned)(nb * RB), 256, lds_lists, s>>>(
                        d.gv, list + b0, nb, dcap, RB, base_id, bm_words, d.clq_kappa4[c - 2], d.clq_probe, sl.rows, stride, cnt);
                }
            } else {
                if (lds_build > 48 * 1024)
                    HIP_TRY(hipFuncSetAttribute((const void *)k_clq_build<K, STATS, ALL_HUB>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_build));
                k_clq_build<K, STATS, ALL_HUB><<<(unsigned)(nb * (dcap / BIG_ROWS_PER_WG)), 256, lds_build, s>>>(
                    d.gv, list + b0, nb, dcap, bias, sl.rows, stride, cnt, byt, byt3);
            }
            HIP_TRY(hipGetLastError());
            if ((rc = timed_end(d, s))) return rc;
            if (K == 4) {
                if ((rc = timed_begin(d, GPM_KIND_CLQ_COUNT, nb, s, c))) return rc;
                if (wide) {
                    HIP_TRY(hipFuncSetAttribute((const void *)k_clq_count4<2, 16>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_count));
                    k_clq_count4<2, 16><<<(unsigned)(nb * phases * seg), 1024, lds_count, s>>>(d.gv, list + b0, nb, dcap,
                                                                                                vt / 32, d.count4_probe, sl.rows, stride, cnt);
                } else {
                    HIP_TRY(hipFuncSetAttribute((const void *)k_clq_count4<4, 8>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_count));
                    k_clq_count4<4, 8><<<(unsigned)(nb * phases * seg), 512, lds_count, s>>>(d.gv, list + b0, nb, dcap,
                                                                                              vt / 32, d.count4_probe, sl.rows, stride, cnt);
                }
                HIP_TRY(hipGetLastError());
                if ((rc = timed_end(d, s))) return rc;
            }
        }
    }
    return GPM_OK;
}

template <int K, bool STATS>
static int clique_launch(DevState &d, ChunkSlot &sl, const uint32_t n[CLQ_CLASSES], bool all_hub, int64_t chunk_hi) {
    hipStream_t s = sl.stream;
    unsigned long long *cnt = d.clq_slots, *byt = d.clq_slots + (size_t)GPM_KIND_CLIQUE * CLQ_SLOTS;
    const int32_t *l = sl.lists;
    const int64_t st = d.clq_stride;
    const int bias = d.clq_probe_bias;
    int rc;
    if constexpr (K == 3 || K == 4) {
        // the gather-only build kernel needs every universe inside the hub matrix and no list-walk bias
        const bool fast = all_hub && bias == 0;
        rc = fast ? clique_launch_big<K, STATS, true>(d, sl, n, bias, chunk_hi)
                  : clique_launch_big<K, STATS, false>(d, sl, n, bias, chunk_hi);
        if (rc) return rc;
    } else {
        if (n[4]) {  // rows in global scratch, persistent workgroups
            unsigned blocks = (unsigned)std::min<uint64_t>(n[4], (uint64_t)d.clq_scratch_blocks);
            if ((rc = timed_begin(d, GPM_KIND_CLIQUE, (int64_t)n[4], s, 4))) return rc;
            k_clique<K, 256, false, STATS><<<blocks, 256, clq_lds_bytes(K, 256, CLQ_MAX_D, false), s>>>(
                d.gv, l + 4 * st, (int64_t)n[4], CLQ_MAX_D, bias, sl.scratch, cnt, byt);
            HIP_TRY(hipGetLastError());
            if ((rc = timed_end(d, s))) return rc;
        }
        for (int c = 3; c >= 2; --c) {
            if (!n[c]) continue;
            const size_t lds = clq_lds_bytes(K, 512, 1024, true);
            HIP_TRY(hipFuncSetAttribute((const void *)k_clique<K, 512, true, STATS>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if ((rc = timed_begin(d, GPM_KIND_CLIQUE, (int64_t)n[c], s, c))) return rc;
            k_clique<K, 512, true, STATS><<<(unsigned)n[c], 512, lds, s>>>(d.gv, l + c * st, (int64_t)n[c], 1024, bias,
                                                                    nullptr, cnt, byt);
            HIP_TRY(hipGetLastError());
            if ((rc = timed_end(d, s))) return rc;
        }
    }
    if (n[1]) {
        if ((rc = timed_begin(d, GPM_KIND_CLIQUE, (int64_t)n[1], s, 1))) return rc;
        k_clique<K, 256, true, STATS><<<(unsigned)n[1], 256, clq_lds_bytes(K, 256, 256, true), s>>>(
            d.gv, l + st, (int64_t)n[1], 256, bias, nullptr, cnt, byt);
        HIP_TRY(hipGetLastError());
        if ((rc = timed_end(d, s))) return rc;
    }
    if constexpr (K == 3 || K == 4) {
        // universes of at most 64, 32, 16 vertices: a group of that many lanes per root, rows in registers
#define TINY_LAUNCH(CLS, GG)                                                                                          \
    if (n[CLS]) {                                                                                                     \
        if ((rc = timed_begin(d, GPM_KIND_CLIQUE, (int64_t)n[CLS], s, CLS))) return rc;                               \
        const unsigned blocks = (unsigned)(((uint64_t)n[CLS] + 256 / GG - 1) / (256 / GG));                          \
        k_clique_tiny<K, GG, STATS><<<blocks, 256, 0, s>>>(d.gv, l + (int64_t)(CLS) * st, (int64_t)n[CLS], cnt, byt); \
        HIP_TRY(hipGetLastError());                                                                                   \
        if ((rc = timed_end(d, s))) return rc;                                                                        \
    }
        TINY_LAUNCH(CLQ_TINY + 2, 64)
        TINY_LAUNCH(CLQ_TINY + 1, 32)
        TINY_LAUNCH(CLQ_TINY, 16)
#undef TINY_LAUNCH
    }
    if (n[0]) {
        if ((rc = timed_begin(d, GPM_KIND_CLIQUE, (int64_t)n[0], s, 0))) return rc;
        k_clique<K, 64, true, STATS><<<(unsigned)n[0], 64, clq_lds_bytes(K, 64, 64, true), s>>>(d.gv, l, (int64_t)n[0], 64, bias,
                                                                                     nullptr, cnt, byt);
        HIP_TRY(hipGetLastError());
        if ((rc = timed_end(d, s))) return rc;
    }
    return GPM_OK;
}

static int expand(const SearchArgs &a, DevState &d, int k, int64_t nrows);

// Largest universe the bitmap kernels take for a K-clique (larger roots go through the generic path).
static int clique_max_d(const gpm_ctx *ctx) {
    const int64_t cap = ctx->clique_k <= 4 ? CLQ_BIG_MAX_D : CLQ_MAX_D;
    return (int)(ctx->clique_max_d_env < 0 ? cap : std::min<int64_t>(cap, ctx->clique_max_d_env));
}

// Largest universe of the in-LDS kernel's classes 0 and 1.  GPM_CLIQUE_SMALL_MAX lowers it for triangles and
// 4-cliques (which have the build / count pair for everything larger), so that tests on small graphs run
// those kernels.
static int clique_small_max(const gpm_ctx *ctx) { return ctx->clique_k > 4 ? 256 : ctx->clique_small_max; }

// Largest universe of the lane-group kernel (triangles and 4-cliques), 0 when it is not used.
static int clique_tiny_max(const gpm_ctx *ctx) {
    return ctx->clique_k <= 4 ? std::min(ctx->clique_tiny, clique_small_max(ctx)) : 0;
}

// One chunk of roots through the bitmap kernel, enqueued on the slot's stream
// without any host round trip: the per-class root counts come from the chunk
// table.  Roots whose N+ exceeds the kernel's universe (class 4) go through
// the generic level-by-level path instead.
static int clique_chunk(const SearchArgs &a, DevState &d, ChunkSlot &sl, const Chunk &ch) {
    gpm_ctx *ctx = a.ctx;
    hipStream_t s = sl.stream;
    HIP_TRY(hipMemsetAsync(sl.counts, 0, 16 * sizeof(unsigned long long), s));
    k_clique_bin<<<grid_for(ch.hi - ch.lo), 256, 0, s>>>(d.gv, ctx->clique_k, clique_max_d(ctx), clique_small_max(ctx), clique_tiny_max(ctx), ch.lo, ch.hi, sl.lists,
                                                         d.clq_stride, sl.counts);
    HIP_TRY(hipGetLastError());
    d.rows_in[0] += (uint64_t)ch.n[0] + ch.n[1] + ch.n[2] + ch.n[3] + ch.n[4] + ch.n[5] + ch.n[CLQ_TINY] +
                    ch.n[CLQ_TINY + 1] + ch.n[CLQ_TINY + 2];
    int rc = GPM_OK;
    // roots run from vertex n-1-lo down to n-hi; universes only hold higher vertices
    const bool all_hub = d.gv.hub != nullptr && (int64_t)ctx->info.n - ch.hi >= (int64_t)d.gv.hub_base;
    switch (ctx->clique_k) {
        case 3: rc = ctx->accounting ? clique_launch<3, true>(d, sl, ch.n, all_hub, ch.hi) : clique_launch<3, false>(d, sl, ch.n, all_hub, ch.hi); break;
        case 4: rc = ctx->accounting ? clique_launch<4, true>(d, sl, ch.n, all_hub, ch.hi) : clique_launch<4, false>(d, sl, ch.n, all_hub, ch.hi); break;
        case 5: rc = ctx->accounting ? clique_launch<5, true>(d, sl, ch.n, all_hub, ch.hi) : clique_launch<5, false>(d, sl, ch.n, all_hub, ch.hi); break;
        case 6: rc = ctx->accounting ? clique_launch<6, true>(d, sl, ch.n, all_hub, ch.hi) : clique_launch<6, false>(d, sl, ch.n, all_hub, ch.hi); break;
        case 7: rc = ctx->accounting ? clique_launch<7, true>(d, sl, ch.n, all_hub, ch.hi) : clique_launch<7, false>(d, sl, ch.n, all_hub, ch.hi); break;
        case 8: rc = ctx->accounting ? clique_launch<8, true>(d, sl, ch.n, all_hub, ch.hi) : clique_launch<8, false>(d, sl, ch.n, all_hub, ch.hi); break;
        case 9: rc = ctx->accounting ? clique_launch<9, true>(d, sl, ch.n, all_hub, ch.hi) : clique_launch<9, false>(d, sl, ch.n, all_hub, ch.hi); break;
        case 10: rc = ctx->accounting ? clique_launch<10, true>(d, sl, ch.n, all_hub, ch.hi) : clique_launch<10, false>(d, sl, ch.n, all_hub, ch.hi); break;
        default: return fail(GPM_ERUNTIME, "clique size %d not supported", ctx->clique_k);
    }
    if (rc) return rc;
    const int generic = CLQ_GENERIC;  // roots too large for the bitmap kernels
    if (ch.n[generic]) {
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpyAsync(d.F[0], sl.lists + (int64_t)generic * d.clq_stride,
                               (size_t)ch.n[generic] * sizeof(int32_t), hipMemcpyDeviceToDevice, d.stream));
        rc = expand(a, d, 1, (int64_t)ch.n[generic]);
    }
    return rc;
}

// Buffers of the 4-cycle path on one device: the two-path prefix over the arcs,
// the two-path count of every root, the counter arrays and the root queues.
static int ensure_cycle4(gpm_ctx *ctx, DevState &d) {
    if (d.cyc_cost) return GPM_OK;
    HIP_TRY(hipSetDevice(d.dev));
    hipStream_t s = d.stream;
    const int64_t n = ctx->info.n, m = d.m[0];
    d.cyc_stride = (n + 63) / 64 * 64 + 64;
    const int64_t budget = env_i64("GPM_CYCLE4_MEM_GB", 32) << 30;
    // half of the budget for the heavy roots' arrays (CYC_HEAVY_GROUPS groups of cyc_batch), the rest for
    // the light workgroups' arrays
    const int64_t arrays = std::max<int64_t>(8, budget / (4 * d.cyc_stride));
    d.cyc_batch = (int)std::min<int64_t>(CYC_BATCH, std::max<int64_t>(1, arrays / 2 / CYC_HEAVY_GROUPS));
    d.cyc_blocks = (int)std::min<int64_t>(env_i64("GPM_CYC_LIGHT_BLOCKS", 2048),
                                          std::max<int64_t>(64, arrays - CYC_HEAVY_GROUPS * d.cyc_batch));
    const size_t dense_bytes = ((size_t)d.cyc_blocks + CYC_HEAVY_GROUPS * d.cyc_batch) * (size_t)d.cyc_stride * sizeof(uint32_t);
    HIP_TRY(hipMalloc(&d.cyc_cost, ((size_t)n + 1) * sizeof(int64_t)));
    HIP_TRY(hipMalloc(&d.cyc_wp, ((size_t)m + 2) * sizeof(int64_t)));
    HIP_TRY(hipMalloc(&d.cyc_dense, dense_bytes));
    HIP_TRY(hipMalloc(&d.cyc_queue, 1024 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d.cyc_dense, 0, dense_bytes, s));
    HIP_TRY(hipMemsetAsync(d.cyc_wp, 0, sizeof(int64_t), s));
    if (m > 0) {
        Scratch len_buf, tmp;
        size_t tmp_bytes = 0;
        HIP_TRY(len_buf.alloc((size_t)m * sizeof(int64_t)));
        int64_t *len = len_buf.as<int64_t>();
        k_cycle4_arc_paths<<<grid_for(m), 256, 0, s>>>(d.gv, m, len);
        HIP_TRY(hipGetLastError());
        HIP_TRY(rocprim::inclusive_scan(nullptr, tmp_bytes, len, d.cyc_wp + 1, (size_t)m, rocprim::plus<int64_t>(), s));
        HIP_TRY(tmp.alloc(tmp_bytes));
        HIP_TRY(rocprim::inclusive_scan(tmp.p, tmp_bytes, len, d.cyc_wp + 1, (size_t)m, rocprim::plus<int64_t>(), s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    HIP_TRY(hipFuncSetAttribute((const void *)k_cycle4_heavy, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)CYC_HEAVY_LDS_BYTES));
    HIP_TRY(hipFuncSetAttribute((const void *)k_cycle4_ranges, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)CYC_RANGE_LDS_BYTES(CYC_RANGE_MAX_IDS)));
    if (n > 0) k_cycle4_cost<<<grid_for(n), 256, 0, s>>>(d.gv, d.cyc_wp, n, d.cyc_cost);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return GPM_OK;
}

// Buffers of the octahedron path on one device, sized for the graph: the kernel needs the longest N+ to fit
// its LDS rows and (longest N+ / 32 + 1) words per vertex and workgroup for the masks.  octa_wgs = -1 when
// the graph does not fit (the search then takes the general kernels).
static int ensure_octa(gpm_ctx *ctx, DevState &d) {
    if (d.octa_wgs) return GPM_OK;
    HIP_TRY(hipSetDevice(d.dev));
    hipStream_t s = d.stream;
    const int64_t n = ctx->info.n;
    Scratch mx;
    HIP_TRY(mx.alloc(sizeof(int)));
    HIP_TRY(hipMemsetAsync(mx.p, 0, sizeof(int), s));
    if (n > 0) k_max_plus<<<grid_for(n), 256, 0, s>>>(d.gv, mx.as<int>());
    HIP_TRY(hipGetLastError());
    int max_plus = 0;
    HIP_TRY(hipMemcpyAsync(&max_plus, mx.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    d.octa_wgs = -1;
    if (max_plus > OCTA_MAX_D) return GPM_OK;
    const int W = std::max(1, (max_plus + 31) / 32);
    const int64_t budget = env_i64("GPM_OCTA_MEM_GB", 32) << 30;
    const int64_t mask_stride = (W + 2) & ~1;  // as the kernels lay the masks out
    const int64_t per_wg = n * (mask_stride + 1) * 4;
    // slabs: one per root of a round
    const int64_t wgs = std::min<int64_t>(env_i64("GPM_OCTA_BLOCKS", 2048), per_wg > 0 ? budget / per_wg : 0);
    if (wgs < 64) return GPM_OK;
    const size_t mask_bytes = (size_t)wgs * (size_t)n * (size_t)mask_stride * sizeof(uint32_t);
    const size_t rows_bytes = (size_t)wgs * (size_t)(32 * W) * (size_t)OCTA_ROW_WORDS(W) * sizeof(uint32_t);
    if (hipMalloc(&d.octa_mask, mask_bytes) != hipSuccess || hipMalloc(&d.octa_touched, (size_t)wgs * (size_t)n * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&d.octa_rows, rows_bytes) != hipSuccess || hipMalloc(&d.octa_hdr, (size_t)wgs * sizeof(OctaHeader)) != hipSuccess ||
        hipMalloc(&d.octa_queue, 1024 * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipGetLastError();
        dev_free(d.octa_mask);
        dev_free(d.octa_touched);
        dev_free(d.octa_rows);
        dev_free(d.octa_hdr);
        dev_free(d.octa_queue);
        d.octa_mask = nullptr;
        d.octa_touched = nullptr;
        d.octa_rows = nullptr;
        d.octa_hdr = nullptr;
        d.octa_queue = nullptr;
        return GPM_OK;
    }
    HIP_TRY(hipMemsetAsync(d.octa_mask, 0, mask_bytes, s));
    HIP_TRY(hipStreamSynchronize(s));
    d.octa_w = W;
    d.octa_wgs = (int)wgs;
    return GPM_OK;
}

// Cut the root range into chunks of similar estimated cost (heaviest roots
// first, so early chunks are short), capped at chunk_roots roots each.  Every
// context over the same graph, plan, root range and hint builds the same table,
// which is what lets several processes share one cursor.
static int prepare_chunks(gpm_ctx *ctx) {
    DevState &d = ctx->devs[0];
    HIP_TRY(hipSetDevice(d.dev));
    hipStream_t s = d.stream;
    ctx->chunks.clear();
    const int64_t lo = ctx->root_lo, hi = ctx->root_hi, nroots = hi - lo;
    if (nroots <= 0) {
        ctx->chunks_valid = true;
        return GPM_OK;
    }
    const int P = ctx->pv.P;
    const bool oriented = P > 1 && ctx->pv.lv[1].src_part == PART_ABOVE;
    const int min_d = ctx->clique_k && ctx->use_clique ? ctx->clique_k - 1 : 0;
    const bool cyc = ctx->cycle4 && ctx->use_cycle4;
    Scratch cost_buf, pref_buf, tmp_buf;
    HIP_TRY(cost_buf.alloc((size_t)nroots * sizeof(int64_t)));
    HIP_TRY(pref_buf.alloc((size_t)nroots * sizeof(int64_t)));
    int64_t *cost = cost_buf.as<int64_t>(), *pref = pref_buf.as<int64_t>();
    if (cyc) {
        int rc = ensure_cycle4(ctx, d);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(cost, d.cyc_cost + lo, (size_t)nroots * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    } else {
        k_root_cost<<<grid_for(nroots), 256, 0, s>>>(d.gv, oriented, min_d, lo, hi, cost);
    }
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::inclusive_scan(nullptr, tmp_bytes, cost, pref, (size_t)nroots, rocprim::plus<int64_t>(), s));
    HIP_TRY(tmp_buf.alloc(tmp_bytes));
    HIP_TRY(rocprim::inclusive_scan(tmp_buf.p, tmp_bytes, cost, pref, (size_t)nroots, rocprim::plus<int64_t>(), s));
    std::vector<int64_t> hpref((size_t)nroots);
    HIP_TRY(hipMemcpyAsync(hpref.data(), pref, (size_t)nroots * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    tmp_buf.release();
    cost_buf.release();
    pref_buf.release();
    ctx->cyc_heavy_roots.clear();
    ctx->cyc_heavy_paths.clear();
    if (cyc)
        for (int64_t i = 0; i < nroots; ++i) {
            const int64_t paths = hpref[(size_t)i] - (i ? hpref[(size_t)i - 1] : 0) - 1;
            if (cyc_is_heavy(ctx->cyc_rule, paths, (int32_t)(ctx->info.n - 1 - (lo + i)))) {
                ctx->cyc_heavy_roots.push_back(lo + i);
                ctx->cyc_heavy_paths.push_back(paths);
            }
        }
    ctx->cyc_heavy_arcs.assign(ctx->cyc_heavy_roots.size(), 0);
    if (!ctx->cyc_heavy_roots.empty()) {
        const size_t nh = ctx->cyc_heavy_roots.size();
        Scratch idx_buf, arcs_buf;
        HIP_TRY(idx_buf.alloc(nh * sizeof(int64_t)));
        HIP_TRY(arcs_buf.alloc(nh * sizeof(int64_t)));
        HIP_TRY(hipMemcpyAsync(idx_buf.p, ctx->cyc_heavy_roots.data(), nh * sizeof(int64_t), hipMemcpyHostToDevice, s));
        k_cycle4_root_arcs<<<grid_for((int64_t)nh), 256, 0, s>>>(d.gv, idx_buf.as<int64_t>(), (int64_t)nh, arcs_buf.as<int64_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ctx->cyc_heavy_arcs.data(), arcs_buf.p, nh * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    ctx->cyc_single_roots.clear();
    ctx->cyc_single_v1.clear();
    ctx->cyc_single_ver++;
    {
        std::vector<size_t> single;  // positions in the heavy lists
        for (size_t h = 0; h < ctx->cyc_heavy_roots.size(); ++h)
            if (cyc_is_ranged(ctx->cyc_rule, ctx->cyc_heavy_paths[h], (int32_t)(ctx->info.n - 1 - ctx->cyc_heavy_roots[h])) &&
                ctx->cyc_heavy_arcs[h] <= (int64_t)CYC_HEAVY_WAVES * GPM_WAVE * ctx->cyc_range_tiles)
                single.push_back(h);
        // heaviest first: a launch takes every K-th root of this list and the GPU starts its workgroups in order
        std::stable_sort(single.begin(), single.end(),
                         [&](size_t a, size_t b) { return ctx->cyc_heavy_paths[a] > ctx->cyc_heavy_paths[b]; });
        for (size_t h : single) {
            ctx->cyc_single_roots.push_back(ctx->cyc_heavy_roots[h]);
            ctx->cyc_single_v1.push_back((int32_t)(ctx->info.n - 1 - ctx->cyc_heavy_roots[h]));
        }
    }
    int64_t want = ctx->chunk_hint > 0 ? ctx->chunk_hint : 32 * (int64_t)ctx->devs.size();
    // the two-path counts are exact costs and every chunk is one persistent launch: few, large chunks
    // chunks of the octahedron path (its own table, below): launches overlap on streams, a quarter of the hint
    const int64_t want_octa = std::max<int64_t>(1, want / 4);
    if (cyc || (ctx->octa && ctx->use_octa)) want = std::max<int64_t>(1, want / 8);
    // the level-by-level path synchronises with the host several times per chunk: fewer, larger chunks
    else if (!(ctx->clique_k && ctx->use_clique)) want = std::max<int64_t>(1, want / 8);
    // A chunk costs the host a handful of enqueues (microseconds each) whatever it holds: a graph whose whole
    // root set is less work than one launch deserves gets one chunk, not the hinted number.
    want = std::min<int64_t>(want, std::max<int64_t>(1, hpref.back() / ctx->min_chunk_cost));
    const int64_t target = std::max<int64_t>(1, hpref.back() / want);
    int64_t begin = 0;
    while (begin < nroots) {
        const int64_t base = begin ? hpref[(size_t)begin - 1] : 0;
        // first index whose prefix reaches base + target
        int64_t end = std::lower_bound(hpref.begin() + begin, hpref.end(), base + target) - hpref.begin() + 1;
        end = std::min({end, nroots, begin + ctx->chunk_roots});
        Chunk c{};
        c.lo = lo + begin;
        c.hi = lo + end;
        ctx->chunks.push_back(c);
        begin = end;
    }
    // Octahedra: one workgroup works through a root, and the heaviest roots take milliseconds.  The roots are
    // therefore sorted by estimated work, heaviest first, and the chunks are ranges of that order: a launch
    // begins with its longest roots and ends with its shortest, so its tail is short, and launches of several
    // chunks overlap on streams of their own.
    ctx->octa_order.clear();
    ctx->octa_chunks.clear();
    ctx->octa_order_ver++;
    if (ctx->octa && ctx->use_octa) {
        bool fits = true;
        for (auto &dv : ctx->devs) {
            int orc = ensure_octa(ctx, dv);
            if (orc) return orc;
            fits = fits && dv.octa_wgs > 0;
        }
        HIP_TRY(hipSetDevice(d.dev));
        if (fits) {
            Scratch k1, k2, v1, v2, tmp, pf;
            HIP_TRY(k1.alloc((size_t)nroots * sizeof(uint64_t)));
            HIP_TRY(k2.alloc((size_t)nroots * sizeof(uint64_t)));
            HIP_TRY(v1.alloc((size_t)nroots * sizeof(int32_t)));
            HIP_TRY(v2.alloc((size_t)nroots * sizeof(int32_t)));
            HIP_TRY(pf.alloc((size_t)nroots * sizeof(uint64_t)));
            k_octa_cost<<<grid_for(nroots), 256, 0, s>>>(d.gv, lo, hi, k1.as<uint64_t>(), v1.as<int32_t>());
            HIP_TRY(hipGetLastError());
            size_t bytes = 0;
            HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, bytes, k1.as<uint64_t>(), k2.as<uint64_t>(), v1.as<int32_t>(),
                                                   v2.as<int32_t>(), (size_t)nroots, 0u, 64u, s));
            HIP_TRY(tmp.alloc(bytes));
            HIP_TRY(rocprim::radix_sort_pairs_desc(tmp.p, bytes, k1.as<uint64_t>(), k2.as<uint64_t>(), v1.as<int32_t>(),
                                                   v2.as<int32_t>(), (size_t)nroots, 0u, 64u, s));
            bytes = 0;
            HIP_TRY(rocprim::inclusive_scan(nullptr, bytes, k2.as<uint64_t>(), pf.as<uint64_t>(), (size_t)nroots,
                                            rocprim::plus<uint64_t>(), s));
            HIP_TRY(tmp.alloc(bytes));
            HIP_TRY(rocprim::inclusive_scan(tmp.p, bytes, k2.as<uint64_t>(), pf.as<uint64_t>(), (size_t)nroots,
                                            rocprim::plus<uint64_t>(), s));
            std::vector<uint64_t> opref((size_t)nroots);
            ctx->octa_order.resize((size_t)nroots);
            HIP_TRY(hipMemcpyAsync(opref.data(), pf.p, (size_t)nroots * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(ctx->octa_order.data(), v2.p, (size_t)nroots * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            // roots with fewer than four higher neighbours hold no copy (estimate 1): the order ends before them
            int64_t active = nroots;
            while (active > 0 && opref[(size_t)active - 1] - (active > 1 ? opref[(size_t)active - 2] : 0) <= 1) --active;
            opref.resize((size_t)active);
            ctx->octa_order.resize((size_t)std::max<int64_t>(active, 1));
            if (active == 0) {  // one empty chunk keeps the table in use
                Chunk c{};
                ctx->octa_chunks.push_back(c);
            }
            const uint64_t per_chunk = std::max<uint64_t>(1, (active ? opref.back() : 1) / (uint64_t)want_octa);
            int64_t ob = 0;
            while (ob < active) {
                const uint64_t base = ob ? opref[(size_t)ob - 1] : 0;
                int64_t oe = std::lower_bound(opref.begin() + ob, opref.end(), base + per_chunk) - opref.begin() + 1;
                oe = std::min(oe, active);
                Chunk c{};
                c.lo = ob;
                c.hi = oe;
                ctx->octa_chunks.push_back(c);
                ob = oe;
            }
        }
    }
    if (ctx->clique_k && ctx->use_clique) {
        const int nch = (int)ctx->chunks.size();
        std::vector<int64_t> bounds((size_t)nch);
        for (int i = 0; i < nch; ++i) bounds[(size_t)i] = ctx->chunks[(size_t)i].lo;
        Scratch dbounds_buf, table_buf;
        HIP_TRY(dbounds_buf.alloc((size_t)nch * sizeof(int64_t)));
        HIP_TRY(table_buf.alloc((size_t)nch * CLQ_CLASSES * sizeof(unsigned int)));
        int64_t *dbounds = dbounds_buf.as<int64_t>();
        unsigned int *table = table_buf.as<unsigned int>();
        HIP_TRY(hipMemcpyAsync(dbounds, bounds.data(), (size_t)nch * sizeof(int64_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(table, 0, (size_t)nch * CLQ_CLASSES * sizeof(unsigned int), s));
        k_chunk_classes<<<grid_for(nroots), 256, 0, s>>>(d.gv, ctx->clique_k, clique_max_d(ctx), clique_small_max(ctx), clique_tiny_max(ctx), lo, hi, dbounds, nch,
                                                         table);
        std::vector<unsigned int> htable((size_t)nch * CLQ_CLASSES);
        HIP_TRY(hipMemcpyAsync(htable.data(), table, htable.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int i = 0; i < nch; ++i)
            for (int c = 0; c < CLQ_CLASSES; ++c)
                ctx->chunks[(size_t)i].n[c] = htable[(size_t)i * CLQ_CLASSES + (size_t)c];
    }
    ctx->chunks_valid = true;
    return GPM_OK;
}

static uint64_t claim_chunk(gpm_ctx *ctx) {
    // GPM_EMULATE_RANKS=N with GPM_EMULATE_RANK=r: take every N-th chunk starting at r, the share one of
    // N processes would get from a shared cursor (for sizing chunk counts on a single GPU)
    static const uint64_t emu_n = (uint64_t)std::max<int64_t>(1, env_i64("GPM_EMULATE_RANKS", 1));
    static const uint64_t emu_r = (uint64_t)env_i64("GPM_EMULATE_RANK", 0) % emu_n;
    uint64_t c = ctx->shared_cursor ? __atomic_fetch_add((uint64_t *)ctx->shared_cursor, 1, __ATOMIC_RELAXED)
                                    : ctx->own_cursor.fetch_add(1, std::memory_order_relaxed);
    return c * emu_n + emu_r;
}

static int device_search(const SearchArgs &a, DevState &d) {
    gpm_ctx *ctx = a.ctx;
    const int P = ctx->pv.P;
    HIP_TRY(hipSetDevice(d.dev));
    int rc = GPM_OK;
    hipStream_t s = d.stream;
    if (d.fixed) {  // the fixed array keeps the last call's contents: what must start at zero is cleared here
        for (int k = 0; k + 1 < std::max(P, 2); ++k) HIP_TRY(hipMemsetAsync(d.pref[k], 0, sizeof(int64_t), s));
    } else if ((rc = ensure_buffers(ctx, d, a.count_only))) {
        return rc;
    }
    HIP_TRY(hipMemsetAsync(d.ctr, 0, CTR_COUNT * sizeof(unsigned long long), s));
    d.canonical = d.alg_bytes = d.tail_bytes = d.chunks = 0;
    for (int kd = 0; kd < GPM_KERNEL_KINDS; ++kd) {
        d.ms_kind[kd] = 0;
        d.launches_kind[kd] = d.bytes_kind[kd] = 0;
    }
    d.events_used = 0;
    memset(d.rows_in, 0, sizeof d.rows_in);
    uint64_t p1_rows = 0;
    const bool clique = a.count_only && ctx->clique_k >= 3 && ctx->use_clique && !a.fixed_memory;
    if (clique) HIP_TRY(hipMemsetAsync(d.clq_slots, 0, GPM_KERNEL_KINDS * CLQ_SLOTS * sizeof(unsigned long long), s));
    HIP_TRY(hipStreamSynchronize(s));  // counters are zero before any slot stream adds to them
    // the octahedron kernel runs when every device of the context could take the graph (prepare_chunks then
    // made the sorted root order and its chunk table)
    const bool octa = a.count_only && ctx->octa && ctx->use_octa && !a.fixed_memory && !ctx->octa_chunks.empty() &&
                      d.octa_wgs > 0;
    if (octa) {
        if (d.octa_order_ver != ctx->octa_order_ver) {
            dev_free(d.octa_order);
            d.octa_order = nullptr;
            HIP_TRY(hipMalloc(&d.octa_order, ctx->octa_order.size() * sizeof(int32_t)));
            HIP_TRY(hipMemcpyAsync(d.octa_order, ctx->octa_order.data(), ctx->octa_order.size() * sizeof(int32_t),
                                   hipMemcpyHostToDevice, s));
            d.octa_order_ver = ctx->octa_order_ver;
        }
        HIP_TRY(hipMemsetAsync(d.octa_queue, 0, 1024 * sizeof(unsigned long long), s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    const std::vector<Chunk> &table = octa ? ctx->octa_chunks : ctx->chunks;
    const bool cyc = a.count_only && ctx->cycle4 && ctx->use_cycle4 && !a.fixed_memory;
    if (cyc) {
        if ((rc = ensure_cycle4(ctx, d))) return rc;
        if (d.cyc_single_ver != ctx->cyc_single_ver) {
            dev_free(d.cyc_single);
            d.cyc_single = nullptr;
            const size_t bytes = ctx->cyc_single_v1.size() * sizeof(int32_t);
            HIP_TRY(hipMalloc(&d.cyc_single, bytes ? bytes : 8));
            if (bytes) HIP_TRY(hipMemcpyAsync(d.cyc_single, ctx->cyc_single_v1.data(), bytes, hipMemcpyHostToDevice, s));
            d.cyc_single_ver = ctx->cyc_single_ver;
        }
        HIP_TRY(hipMemsetAsync(d.cyc_queue, 0, 1024 * sizeof(unsigned long long), s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    // 4-cycles: the heavy roots whose arcs fit one task are worked off apart from the chunks, in `pieces`
    // launches that come first in the cursor's numbering; piece p takes every pieces-th root of the list
    // (heaviest first), so each launch holds thousands of workgroups and begins with its longest roots
    const uint64_t pieces = cyc && ctx->cyc_rule.S && !ctx->cyc_single_v1.empty()
                                ? std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)env_i64("GPM_CYC_PIECES", (int64_t)ctx->chunks.size() / 4), ctx->cyc_single_v1.size()))
                                : 0;
    const uint64_t cyc_ahead = (uint64_t)std::min<int64_t>(CLQ_INFLIGHT, std::max<int64_t>(1, env_i64("GPM_CYC_INFLIGHT", CLQ_INFLIGHT)));
    uint64_t cyc_items = 0;
    if (cyc && d.ring.empty()) {
        d.ring.resize((size_t)CLQ_INFLIGHT * CLQ_INFLIGHT);
        for (auto &ev : d.ring) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    uint64_t seq = 0, cyc_heavy_seq = 0, octa_rounds = 0;
    // chunks in flight on the clique path; 1 runs every launch with the GPU to itself (kernel times
    // then add up to the wall time, which is how a kernel's own rate is measured)
    const int clq_inflight = (int)std::min<int64_t>(CLQ_INFLIGHT, std::max<int64_t>(1, env_i64("GPM_CLIQUE_INFLIGHT", CLQ_INFLIGHT)));
    const auto t_begin = std::chrono::steady_clock::now();
    uint64_t next_fixed = 0;  // fixed-memory search: equal chunks of the root range, no chunk table
    Chunk fixed_chunk{};
    d.ag_rows = 0;
    while (!ctx->stop) {
        if ((rc = gather_check(a, d))) return rc;
        uint64_t c;
        if (a.fixed_memory) {
            c = next_fixed++;
            fixed_chunk.lo = ctx->root_lo + (int64_t)c * d.cap_rows;
            fixed_chunk.hi = std::min(ctx->root_hi, fixed_chunk.lo + d.cap_rows);
            if (fixed_chunk.lo >= ctx->root_hi) break;
        } else {
            c = claim_chunk(ctx);
            if (c >= table.size() + pieces) break;
        }
        if (cyc) {  // at most CLQ_INFLIGHT chunks ahead of the GPU: what is not claimed yet is left for other ranks
            hipEvent_t *ring = d.ring.data() + (cyc_items % cyc_ahead) * CLQ_INFLIGHT;
            if (cyc_items >= cyc_ahead)
                for (int i = 0; i < CLQ_INFLIGHT; ++i) HIP_TRY(hipEventSynchronize(ring[i]));
        }
        if (cyc && c < pieces) {
            const int S = ctx->cyc_rule.S;
            const size_t total = ctx->cyc_single_v1.size();
            const size_t count = (total - (size_t)c + (size_t)pieces - 1) / (size_t)pieces;  // roots c, c + pieces, ...
            hipStream_t hs = d.slots[(int)(cyc_heavy_seq++ % CYC_HEAVY_GROUPS) + 1].stream;
            d.chunks++;
            if ((rc = timed_begin(d, GPM_KIND_CYCLE4, (int64_t)count, hs))) return rc;
            k_cycle4_ranges<<<(unsigned)count, CYC_HEAVY_WAVES * GPM_WAVE, CYC_RANGE_LDS_BYTES(S), hs>>>(
                d.gv, d.cyc_wp, CycBatch{}, d.cyc_single + c, (int)pieces, S, ctx->cyc_range_tiles, nullptr, 0,
                d.ctr + CTR_TAIL_MATCHES, d.ctr + CTR_CYC_BYTES);
            HIP_TRY(hipGetLastError());
            if ((rc = timed_end(d, hs))) return rc;
            hipEvent_t *ring = d.ring.data() + (cyc_items++ % cyc_ahead) * CLQ_INFLIGHT;
            for (int i = 0; i < CLQ_INFLIGHT; ++i) HIP_TRY(hipEventRecord(ring[i], d.slots[i].stream));
            continue;
        }
        const Chunk &ch = a.fixed_memory ? fixed_chunk : table[(size_t)(c - pieces)];
        const int64_t lo = ch.lo, hi = ch.hi;
        d.chunks++;
        if (octa) {
            // A chunk is a range of the sorted root order, worked off in rounds of as many roots as its
            // stream has slabs: k_octa_mask builds the masks of a round's roots, k_octa_sets counts over their
            // touched vertices.  Chunks alternate over octa_streams streams, each with its own share of the
            // slabs, so the rounds of several chunks overlap.
            const int groups = std::max(1, std::min(ctx->octa_streams, d.octa_wgs / 64));
            const int slabs = std::min(1024, d.octa_wgs / groups);
            const int grp = (int)(seq % (uint64_t)groups);
            ChunkSlot &sl = d.slots[grp];
            if (sl.busy) {  // at most `groups` chunks ahead of the GPU: what is not claimed yet is left for other ranks to take
                HIP_TRY(hipEventSynchronize(sl.done));
                sl.busy = false;
            }
            hipStream_t os = sl.stream;
            const size_t mask_stride = (size_t)((d.octa_w + 2) & ~1);
            uint32_t *gmask = d.octa_mask + (size_t)grp * (size_t)slabs * (size_t)d.n * mask_stride;
            int32_t *gtouched = d.octa_touched + (size_t)grp * (size_t)slabs * (size_t)d.n;
            uint32_t *grows = d.octa_rows + (size_t)grp * (size_t)slabs * (size_t)(32 * d.octa_w) * (size_t)OCTA_ROW_WORDS(d.octa_w);
            OctaHeader *ghdr = d.octa_hdr + (size_t)grp * (size_t)slabs;
            d.rows_in[0] += (uint64_t)(hi - lo);
            // a chunk of few roots is a chunk of heavy roots (chunks hold equal estimated work): 16 waves walk
            // the lists of such a root's members
            const bool wide = hi - lo <= ctx->octa_wide_roots;
            const unsigned sets_wgs = (unsigned)std::max(256, 2048 / groups);
            for (int64_t base = lo; base < hi; base += slabs) {
                const int64_t nr = std::min<int64_t>(slabs, hi - base);
                if (octa_rounds && octa_rounds % 1024 == 0) {  // the queue words are used up: all rounds done, then zeroed
                    for (auto &x : d.slots) HIP_TRY(hipStreamSynchronize(x.stream));
                    HIP_TRY(hipMemsetAsync(d.octa_queue, 0, 1024 * sizeof(unsigned long long), s));
                    HIP_TRY(hipStreamSynchronize(s));
                }
                unsigned long long *q = d.octa_queue + octa_rounds % 1024;
                octa_rounds++;
                if ((rc = timed_begin(d, GPM_KIND_OCTA, nr, os, wide ? 16 : 4))) return rc;
                if (wide)
                    k_octa_mask<16><<<(unsigned)nr, 16 * GPM_WAVE, OCTA_MASK_LDS_BYTES(d.octa_w), os>>>(
                        d.gv, d.octa_order, base, base + nr, gmask, gtouched, grows, ghdr, d.octa_w, ctx->octa_seg, d.ctr + CTR_OCTA_BYTES);
                else
                    k_octa_mask<4><<<(unsigned)nr, 4 * GPM_WAVE, OCTA_MASK_LDS_BYTES(d.octa_w), os>>>(
                        d.gv, d.octa_order, base, base + nr, gmask, gtouched, grows, ghdr, d.octa_w, ctx->octa_seg, d.ctr + CTR_OCTA_BYTES);
                HIP_TRY(hipGetLastError());
                if ((rc = timed_end(d, os))) return rc;
                if ((rc = timed_begin(d, GPM_KIND_OCTA, nr, os, 1))) return rc;  // tag 1: the sets kernel
                k_octa_sets<<<sets_wgs, 256, OCTA_SETS_LDS_BYTES(d.octa_w, ctx->octa_seg), os>>>(d.gv, (int)nr, ghdr, gmask, gtouched, grows,
                                                                                 d.octa_w, q, d.ctr + CTR_TAIL_MATCHES);
                HIP_TRY(hipGetLastError());
                if ((rc = timed_end(d, os))) return rc;
            }
            HIP_TRY(hipEventRecord(sl.done, os));
            sl.busy = true;
            seq++;
            if (d.events_used > 4096) {
                for (auto &x : d.slots) HIP_TRY(hipStreamSynchronize(x.stream));
                if ((rc = collect_times(d, ctx->trace))) return rc;
            }
            continue;
        }
        if (cyc) {
            // light roots of the chunk: one launch on the first slot stream (launches share the
            // workgroups' counter arrays, so they run one after the other); heavy roots: one launch
            // and one memset each, spread over CYC_HEAVY_SLABS streams with one counter array each
            hipStream_t ls = d.slots[0].stream;
            if (seq && seq % 1024 == 0) {
                HIP_TRY(hipStreamSynchronize(ls));
                HIP_TRY(hipMemsetAsync(d.cyc_queue, 0, 1024 * sizeof(unsigned long long), ls));
            }
            d.rows_in[0] += (uint64_t)(hi - lo);
            if ((rc = timed_begin(d, GPM_KIND_CYCLE4, hi - lo, ls))) return rc;
            k_cycle4_light<<<(unsigned)d.cyc_blocks, 256, 0, ls>>>(d.gv, d.cyc_wp, lo, hi, d.cyc_cost,
                                                                  ctx->cyc_rule, d.cyc_queue + seq % 1024,
                                                                  d.cyc_dense, d.cyc_stride, d.ctr + CTR_TAIL_MATCHES,
                                                                  d.ctr + CTR_CYC_BYTES);
            HIP_TRY(hipGetLastError());
            if ((rc = timed_end(d, ls))) return rc;
            seq++;
            const auto &hv = ctx->cyc_heavy_roots;
            // heavy roots of the chunk, up to CYC_BATCH per launch; the light launches keep a stream (and
            // with it a hardware queue) of their own, so the batches alternate between the other three.
            // Roots with enough two-paths per range of ids (cyc_is_ranged) count in LDS, one workgroup per
            // block of arcs, S counters at a time; the others are cut into spans of two-paths.
            const int64_t hblock = ctx->cyc_heavy_block, span = ctx->cyc_span;
            const int S = ctx->cyc_rule.S;
            const size_t h_lo = (size_t)(std::lower_bound(hv.begin(), hv.end(), lo) - hv.begin());
            for (int ranged = S ? 1 : 0; ranged >= 0; --ranged) {
                size_t h = h_lo;
                while (h < hv.size() && hv[h] < hi) {
                    CycBatch b{};
                    int64_t max_v1 = 0;
                    bool shared_arrays = !ranged;  // a ranged launch needs the arrays only for roots of several arc blocks
                    for (; b.n < d.cyc_batch && h < hv.size() && hv[h] < hi; ++h) {
                        const int64_t paths = ctx->cyc_heavy_paths[h];
                        const int32_t v1 = (int32_t)(ctx->info.n - 1 - hv[h]);
                        if (cyc_is_ranged(ctx->cyc_rule, paths, v1) != (bool)ranged) continue;
                        const int64_t block_arcs = (int64_t)CYC_HEAVY_WAVES * GPM_WAVE * ctx->cyc_range_tiles;
                        const int64_t tasks = ranged ? (ctx->cyc_heavy_arcs[h] + block_arcs - 1) / block_arcs : (paths + span - 1) / span;
                        if (ranged && tasks == 1) continue;  // in one of the launches of single-task roots
                        if (b.n && (int64_t)b.wg0[b.n] + tasks > (int64_t)1 << 20) break;  // next launch
                        b.v1[b.n] = v1;
                        shared_arrays |= tasks > 1;
                        max_v1 = std::max<int64_t>(max_v1, v1);
                        b.wg0[b.n + 1] = b.wg0[b.n] + (int32_t)tasks;
                        b.n++;
                    }
                    if (!b.n) break;
                    const int group = (int)(cyc_heavy_seq++ % CYC_HEAVY_GROUPS);
                    hipStream_t hs = d.slots[group + 1].stream;
                    uint32_t *dense = d.cyc_dense + ((size_t)d.cyc_blocks + (size_t)group * d.cyc_batch) * (size_t)d.cyc_stride;
                    if ((rc = timed_begin(d, GPM_KIND_CYCLE4, b.n, hs))) return rc;
                    if (ranged)
                        k_cycle4_ranges<<<(unsigned)b.wg0[b.n], CYC_HEAVY_WAVES * GPM_WAVE, CYC_RANGE_LDS_BYTES(S), hs>>>(
                            d.gv, d.cyc_wp, b, nullptr, 1, S, ctx->cyc_range_tiles, dense, d.cyc_stride, d.ctr + CTR_TAIL_MATCHES, d.ctr + CTR_CYC_BYTES);
                    else
                        k_cycle4_heavy<<<(unsigned)b.wg0[b.n], CYC_HEAVY_WAVES * GPM_WAVE, CYC_HEAVY_LDS_BYTES, hs>>>(
                            d.gv, d.cyc_wp, b, dense, d.cyc_stride, span, hblock, d.ctr + CTR_TAIL_MATCHES, d.ctr + CTR_CYC_BYTES);
                    HIP_TRY(hipGetLastError());
                    if (shared_arrays)
                        HIP_TRY(hipMemsetAsync(dense, 0, ((size_t)(b.n - 1) * (size_t)d.cyc_stride + (size_t)max_v1) * sizeof(uint32_t), hs));
                    if ((rc = timed_end(d, hs))) return rc;
                    progress_tick(d.dev, d.chunks, ctx->chunks.size(), 0, (int64_t)h);
                    if (d.events_used > 4096) {
                        for (auto &x : d.slots) HIP_TRY(hipStreamSynchronize(x.stream));
                        HIP_TRY(hipStreamSynchronize(s));
                        if ((rc = collect_times(d, ctx->trace))) return rc;
                    }
                }
            }
            hipEvent_t *ring = d.ring.data() + (cyc_items++ % cyc_ahead) * CLQ_INFLIGHT;
            for (int i = 0; i < CLQ_INFLIGHT; ++i) HIP_TRY(hipEventRecord(ring[i], d.slots[i].stream));
            continue;
        }
        if (clique) {
            ChunkSlot &sl = d.slots[seq++ % clq_inflight];
            if (sl.busy) {  // at most CLQ_INFLIGHT chunks ahead of the GPU
                HIP_TRY(hipEventSynchronize(sl.done));
                sl.busy = false;
            }
            rc = clique_chunk(a, d, sl, ch);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(sl.done, sl.stream));
            sl.busy = true;
            if (d.events_used > 4096) {
                for (auto &x : d.slots) HIP_TRY(hipStreamSynchronize(x.stream));
                if ((rc = collect_times(d, ctx->trace))) return rc;
            }
            continue;
        }
        HIP_TRY(hipMemsetAsync(d.ctr + CTR_LEVEL0, 0, sizeof(unsigned long long), s));
        k_filter_roots<<<grid_for(hi - lo), 256, 0, s>>>(d.gv, ctx->pv.lv[0], lo, hi, d.F[0], d.ctr + CTR_LEVEL0);
        HIP_TRY(hipGetLastError());
        uint64_t n0 = 0;
        rc = read_ctr(d, CTR_LEVEL0, &n0);
        if (rc) return rc;
        if (P == 1) {
            p1_rows += n0;
            d.rows_in[0] += n0;
            if (!a.count_only && n0) {
                for (uint64_t r0 = 0; r0 < n0; r0 += (uint64_t)d.cap_emit) {
                    uint64_t cnt = std::min<uint64_t>((uint64_t)d.cap_emit, n0 - r0);
                    k_emit_roots<<<grid_for((int64_t)cnt), 256, 0, s>>>((int64_t)cnt, d.F[0] + r0, d.perm, d.emit);
                    HIP_TRY(hipGetLastError());
                    if ((rc = hand_over(a, d, cnt))) return rc;
                }
            }
            continue;
        }
        rc = expand(a, d, 1, (int64_t)n0);
        if (rc) return rc;
        if (d.events_used > 4096) {
            HIP_TRY(hipStreamSynchronize(s));
            rc = collect_times(d, ctx->trace);
            if (rc) return rc;
        }
    }
    if (a.gather && !ctx->stop && (rc = gather_finish(a, d))) return rc;
    if (ctx->trace)
        fprintf(stderr, "gpm trace: all chunks enqueued %.3f ms after the search began\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    if (cyc || octa)
        for (auto &x : d.slots) {
            HIP_TRY(hipStreamSynchronize(x.stream));
            x.busy = false;
        }
    if (clique) {
        for (auto &x : d.slots) {
            HIP_TRY(hipStreamSynchronize(x.stream));
            x.busy = false;
        }
        k_sum_slots<<<1, 64, 0, s>>>(d.clq_slots, CLQ_SLOTS, d.ctr + CTR_TAIL_MATCHES);
        for (int kd : {GPM_KIND_CLIQUE, GPM_KIND_CLQ_BUILD, GPM_KIND_CLQ_COUNT})
            k_sum_slots<<<1, 64, 0, s>>>(d.clq_slots + (size_t)kd * CLQ_SLOTS, CLQ_SLOTS, d.ctr + CTR_KIND_BYTES + kd);
    }
    HIP_TRY(hipMemcpyAsync(d.ctr_host, d.ctr, CTR_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    rc = collect_times(d, ctx->trace);
    if (rc) return rc;
    if (a.count_only) d.canonical = d.ctr_host[CTR_MATCHES] + d.ctr_host[CTR_TAIL_MATCHES] + p1_rows;
    else if (P == 1) d.canonical = p1_rows;
    d.bytes_kind[GPM_KIND_TAIL] = d.ctr_host[CTR_TAIL_BYTES];
    d.bytes_kind[GPM_KIND_EXTEND] = d.ctr_host[CTR_BYTES];
    d.bytes_kind[GPM_KIND_CYCLE4] = d.ctr_host[CTR_CYC_BYTES];
    d.bytes_kind[GPM_KIND_OCTA] = d.ctr_host[CTR_OCTA_BYTES];
    for (int kd : {GPM_KIND_CLIQUE, GPM_KIND_CLQ_BUILD, GPM_KIND_CLQ_COUNT})
        d.bytes_kind[kd] = d.ctr_host[CTR_KIND_BYTES + kd];
    d.tail_bytes = 0;
    for (int kd = 0; kd < GPM_KERNEL_KINDS; ++kd)
        if (kd != GPM_KIND_EXTEND) d.tail_bytes += d.bytes_kind[kd];
    d.alg_bytes = d.ctr_host[CTR_BYTES] + d.tail_bytes + 4ull * d.rows_in[0];
    return GPM_OK;
}

static int search_impl(gpm_ctx *ctx, int count_only, gpm_row_sink sink, gpm_text_sink text_sink, void *user,
                       uint64_t *total) {
    if (!ctx) return fail(GPM_EINVAL, "null context");
    if (!ctx->have_graph || !ctx->have_plan) return fail(GPM_ESTATE, "upload a graph and set a plan first");
    if (!count_only && !sink && !text_sink) return fail(GPM_EINVAL, "output mode needs a sink");
    (void)hipGetLastError();
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    ctx->stop = false;
    ctx->own_cursor = 0;
    if (!ctx->chunks_valid) {
        int prc = prepare_chunks(ctx);
        if (prc) return prc;
    }
    // one worker thread per GPU; a single GPU searches on the calling thread, unless it has a communicator
    // (GPM_FORCE_RCCL=1), in which case output mode runs the gather rounds of a multi-GPU context with one worker
    const bool threaded = ctx->devs.size() > 1 || (!ctx->comms.empty() && !count_only);
    SearchArgs a{ctx, count_only != 0, !threaded, sink, user, text_sink};
    a.gather = !count_only && threaded && ctx->gather_on_device;
    ctx->gs.arrived = ctx->gs.finished = ctx->gs.waiting = 0;
    ctx->gs.round_failed = false;
    ctx->gs.requested = false;
    ctx->gs.rounds = ctx->gs.gathered_rows = 0;
    int rc = GPM_OK;
    if (!threaded) {
        rc = device_search(a, ctx->devs[0]);
    } else {
        std::vector<std::thread> threads;
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            ctx->workers_live = (int)ctx->devs.size();
            ctx->queue.clear();
        }
        for (auto &d : ctx->devs) {
            threads.emplace_back([&a, &d, ctx] {
                d.rc = device_search(a, d);
                if (d.rc) {
                    d.err = gpm_hip_last_error();
                    ctx->stop = true;
                }
                std::lock_guard<std::mutex> lk(ctx->mu);
                ctx->workers_live--;
                ctx->cv.notify_all();
            });
        }
        for (;;) {  // the calling thread is the only one that runs the sink
            RowBatch batch;
            {
                std::unique_lock<std::mutex> lk(ctx->mu);
                ctx->cv.wait(lk, [&] { return !ctx->queue.empty() || ctx->workers_live == 0; });
                if (ctx->queue.empty()) break;
                batch = std::move(ctx->queue.front());
                ctx->queue.pop_front();
                ctx->cv.notify_all();
            }
            const bool is_text = text_sink != nullptr;
            if (!ctx->stop && (is_text ? text_sink(user, batch.text.data(), batch.text.size(), batch.nrows) != 0
                                       : (sink && sink(user, batch.rows.data(), batch.nrows, ctx->pv.P) != 0))) {
                ctx->stop = true;
                rc = fail(GPM_ESINK, "the row sink stopped the search");
                ctx->cv.notify_all();
            }
        }
        for (auto &t : threads) t.join();
        for (auto &d : ctx->devs)
            if (d.rc && rc == GPM_OK) rc = fail(d.rc, "%s", d.err.c_str());
    }
    // Sum the per-GPU counts: one all-reduce of a 64-bit word over RCCL when the
    // context has communicators, so that every GPU (and the host, from GPU 0)
    // holds the total.
    uint64_t reduced = 0;
    bool have_reduced = false;
    if (rc == GPM_OK && !ctx->comms.empty()) {
        const Rccl &R = rccl();
        int nrc = 0;
        for (auto &d : ctx->devs) {
            (void)hipSetDevice(d.dev);
            d.ctr_host[CTR_CLQ_CLASS0] = d.canonical;
            (void)hipMemcpyAsync(d.ctr + CTR_CLQ_CLASS0, d.ctr_host + CTR_CLQ_CLASS0, sizeof(unsigned long long),
                                 hipMemcpyHostToDevice, d.stream);
        }
        nrc = R.GroupStart();
        for (size_t i = 0; i < ctx->devs.size() && nrc == 0; ++i) {
            DevState &d = ctx->devs[i];
            (void)hipSetDevice(d.dev);
            nrc = R.AllReduce(d.ctr + CTR_CLQ_CLASS0, d.ctr + CTR_CLQ_CLASS0 + 1, 1, RCCL_UINT64, RCCL_SUM,
                              ctx->comms[i], d.stream);
        }
        if (nrc == 0) nrc = R.GroupEnd();
        if (nrc != 0) {
            rc = fail(GPM_ERUNTIME, "RCCL all-reduce of the counts failed: %s", R.GetErrorString(nrc));
        } else {
            DevState &d0 = ctx->devs[0];
            (void)hipSetDevice(d0.dev);
            hipError_t e = hipMemcpyAsync(d0.ctr_host + CTR_CLQ_CLASS0 + 1, d0.ctr + CTR_CLQ_CLASS0 + 1,
                                          sizeof(unsigned long long), hipMemcpyDeviceToHost, d0.stream);
            for (auto &d : ctx->devs) {
                (void)hipSetDevice(d.dev);
                if (e == hipSuccess) e = hipStreamSynchronize(d.stream);
            }
            if (e != hipSuccess) rc = fail(GPM_ERUNTIME, "reading the reduced count failed: %s", hipGetErrorString(e));
            reduced = d0.ctr_host[CTR_CLQ_CLASS0 + 1];
            have_reduced = rc == GPM_OK;
        }
    }
    gpm_stats &st = ctx->stats;
    memset(&st, 0, sizeof st);
    for (size_t i = 0; i < ctx->devs.size(); ++i) {
        DevState &d = ctx->devs[i];
        st.canonical += d.canonical;
        st.alg_bytes += d.alg_bytes;
        st.tail_alg_bytes += d.tail_bytes;
        for (int kd = 0; kd < GPM_KERNEL_KINDS; ++kd) {
            st.kernel_ms[kd] += d.ms_kind[kd];
            st.kernel_launches[kd] += d.launches_kind[kd];
            st.kernel_alg_bytes[kd] += d.bytes_kind[kd];
            if (kd == GPM_KIND_EXTEND) {
                st.ms_extend_kernel += d.ms_kind[kd];
                st.extend_launches += d.launches_kind[kd];
            } else {
                st.ms_tail_kernel += d.ms_kind[kd];
                st.tail_launches += d.launches_kind[kd];
            }
        }
        for (int k = 0; k < GPM_MAXP; ++k) st.rows_in[k] += d.rows_in[k];
        st.chunks[i] = d.chunks;
    }
    if (have_reduced) {
        if (reduced != st.canonical) rc = fail(GPM_ERUNTIME, "RCCL sum %llu differs from the host sum %llu",
                                               (unsigned long long)reduced, (unsigned long long)st.canonical);
        st.canonical = reduced;
    }
    st.gather_rounds = ctx->gs.rounds;
    st.gathered_rows = ctx->gs.gathered_rows;
    st.matches = st.canonical * ctx->plan.n_aut;
    st.ms_total =
        1e3 * (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0);
    if (total) *total = st.matches;
    return rc;
}

extern "C" int gpm_hip_search(gpm_ctx *ctx, int count_only, gpm_row_sink sink, void *user, uint64_t *total) {
    return search_impl(ctx, count_only, sink, nullptr, user, total);
}

extern "C" int gpm_hip_search_text(gpm_ctx *ctx, gpm_text_sink sink, void *user, uint64_t *total) {
    if (!sink) return fail(GPM_EINVAL, "text output needs a sink");
    return search_impl(ctx, 0, nullptr, sink, user, total);
}

// Count-only search of a single-GPU context that allocates nothing: its
// frontiers, candidate ranges, prefix sums, scan workspace and counters are
// carved out of this GPU's copy of gpm_scratch, whose address gpm_hip_create
// looked up.  The pinned words results are read back through were made by
// gpm_hip_create as well.  Runs the level-by-level and fused-tail kernels;
// every call on one GPU uses the same array, so overlapping calls would race.
extern "C" int gpm_hip_count_noalloc(gpm_ctx *ctx, uint64_t *total) {
    if (!ctx || !total) return fail(GPM_EINVAL, "null argument");
    if (!ctx->have_graph || !ctx->have_plan) return fail(GPM_ESTATE, "upload a graph and set a plan first");
    if (ctx->devs.size() != 1) return fail(GPM_EINVAL, "the no-allocation search runs on a single-GPU context");
    if (!ctx->scratch0) return fail(GPM_ESTATE, "the fixed scratch array was not located");
    if (ctx->shared_cursor)
        return fail(GPM_ESTATE, "the no-allocation search walks every root itself: not on a context that shares a chunk cursor");
    const int64_t cap = NOALLOC_ROWS;
    const int64_t longest = (int64_t)std::max(ctx->info.max_out_deg, ctx->info.max_in_deg);
    if (longest + 64 > cap)
        return fail(GPM_EINVAL,
                    "the longest adjacency list has %lld entries; the no-allocation search takes at most %lld "
                    "(use gpm_hip_search)", (long long)longest, (long long)(cap - 64));
    (void)hipGetLastError();
    const double t0 = now_seconds();
    DevState &d0 = ctx->devs[0];
    DevState &v = ctx->na;
    HIP_TRY(hipSetDevice(d0.dev));
    v.fixed = true;
    v.dev = d0.dev;
    v.stream = d0.stream;
    v.n = d0.n;
    v.gv = d0.gv;
    v.perm = d0.perm;
    v.ls_terms = d0.ls_terms;
    v.ctr_host = d0.ctr_host;
    v.i64_host = d0.i64_host;
    v.cap_rows = cap;
    v.cap_emit = 0;
    // carve the array: every piece starts on a 256-byte boundary
    unsigned char *base = ctx->scratch0;
    size_t used = 0;
    auto take = [&](size_t bytes) {
        unsigned char *p = base + used;
        used += (bytes + 255) & ~(size_t)255;
        return p;
    };
    for (int k = 0; k + 1 < GPM_MAXP; ++k) {
        v.F[k] = (int32_t *)take((size_t)cap * (size_t)(k + 1) * sizeof(int32_t));
        v.beg[k] = (int64_t *)take((size_t)cap * sizeof(int64_t));
        v.cnt[k] = (int64_t *)take((size_t)cap * sizeof(int64_t));
        v.pref[k] = (int64_t *)take(((size_t)cap + 1) * sizeof(int64_t));
        v.wsel[k] = (int8_t *)take((size_t)cap);
    }
    v.over_rows = (int64_t *)take(2 * (size_t)cap * sizeof(int64_t));
    v.ctr = (unsigned long long *)take(CTR_COUNT * sizeof(unsigned long long));
    v.i64_dev = (int64_t *)take(4 * sizeof(int64_t));
    if (used > NOALLOC_BYTES) return fail(GPM_ERUNTIME, "fixed scratch layout needs %zu bytes", used);
    v.scan_tmp = take(0);
    size_t scan_bytes = 0;  // a size query: allocates nothing
    HIP_TRY(rocprim::inclusive_scan(nullptr, scan_bytes, v.cnt[0], v.pref[0] + 1, (size_t)cap, rocprim::plus<int64_t>(),
                                    v.stream));
    v.scan_tmp_bytes = scan_bytes ? scan_bytes : 8;
    if (used + v.scan_tmp_bytes > NOALLOC_BYTES)
        return fail(GPM_ERUNTIME, "fixed scratch layout needs %zu bytes", used + v.scan_tmp_bytes);
    ctx->stop = false;
    SearchArgs a{ctx, true, true, nullptr, nullptr, nullptr, false, true};
    int rc = device_search(a, v);
    if (rc != GPM_OK) {
        (void)hipStreamSynchronize(v.stream);
        return rc;
    }
    gpm_stats &st = ctx->stats;
    memset(&st, 0, sizeof st);
    st.canonical = v.canonical;
    st.matches = v.canonical * ctx->plan.n_aut;
    st.alg_bytes = v.alg_bytes;
    st.tail_alg_bytes = v.tail_bytes;
    for (int k = 0; k < GPM_MAXP; ++k) st.rows_in[k] = v.rows_in[k];
    for (int kd = 0; kd < GPM_KERNEL_KINDS; ++kd) st.kernel_alg_bytes[kd] = v.bytes_kind[kd];
    st.chunks[0] = v.chunks;
    st.ms_total = 1e3 * (now_seconds() - t0);
    *total = st.matches;
    return GPM_OK;
}

extern "C" const char *gpm_hip_kernel_name(int kind) {
    return kind >= 0 && kind < GPM_KERNEL_KINDS ? kKindNames[kind] : "";
}

extern "C" int gpm_hip_stats(const gpm_ctx *ctx, gpm_stats *stats) {
    if (!ctx || !stats) return fail(GPM_EINVAL, "null argument");
    *stats = ctx->stats;
    return GPM_OK;
}
