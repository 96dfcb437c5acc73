"""ctypes binding of the C ABI (include/gpm_hip.h) and of the C host library.

This is plumbing for tests and the benchmark: the product is the C program
gpm-par.exe plus libgpm_hip.so.  Function names follow the reference's
(readGraph / makeGraphOrdering / findMatches, reference: src/graph.c:24-86 and
src/gpm-par.c:245).  There is no CPU fallback: if the HIP library is missing
or no GPU is visible the calls raise.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
MAXP, MAXC, KINDS = 10, 9, 8


class GpmError(RuntimeError):
    pass


class gpm_plan(ctypes.Structure):
    _fields_ = [
        ("n_levels", ctypes.c_int32),
        ("order", ctypes.c_int32 * MAXP),
        ("src_level", ctypes.c_int32 * MAXP),
        ("src_dir", ctypes.c_int32 * MAXP),
        ("n_cons", ctypes.c_int32 * MAXP),
        ("cons_level", (ctypes.c_int32 * MAXC) * MAXP),
        ("cons_dir", (ctypes.c_int32 * MAXC) * MAXP),
        ("n_lower", ctypes.c_int32 * MAXP),
        ("lower_level", (ctypes.c_int32 * MAXC) * MAXP),
        ("n_upper", ctypes.c_int32 * MAXP),
        ("upper_level", (ctypes.c_int32 * MAXC) * MAXP),
        ("label", ctypes.c_int32 * MAXP),
        ("min_out_deg", ctypes.c_int32 * MAXP),
        ("min_in_deg", ctypes.c_int32 * MAXP),
        ("n_aut", ctypes.c_uint64),
        ("aut", ctypes.POINTER(ctypes.c_int8)),
    ]


class gpm_graph_info(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("m_out", ctypes.c_int64), ("symmetric", ctypes.c_int32),
                ("simple", ctypes.c_int32), ("max_out_deg", ctypes.c_int32), ("max_in_deg", ctypes.c_int32),
                ("has_labels", ctypes.c_int32), ("hub_rows", ctypes.c_int32)]


class gpm_stats(ctypes.Structure):
    _fields_ = [("matches", ctypes.c_uint64), ("canonical", ctypes.c_uint64),
                ("rows_in", ctypes.c_uint64 * MAXP), ("alg_bytes", ctypes.c_uint64),
                ("tail_alg_bytes", ctypes.c_uint64), ("ms_total", ctypes.c_double),
                ("ms_tail_kernel", ctypes.c_double), ("ms_extend_kernel", ctypes.c_double),
                ("tail_launches", ctypes.c_uint64), ("extend_launches", ctypes.c_uint64),
                ("chunks", ctypes.c_uint64 * 8),
                ("kernel_ms", ctypes.c_double * KINDS), ("kernel_launches", ctypes.c_uint64 * KINDS),
                ("kernel_alg_bytes", ctypes.c_uint64 * KINDS),
                ("gather_rounds", ctypes.c_uint64), ("gathered_rows", ctypes.c_uint64)]


PLAN_SYMMETRIC_DATA, PLAN_BREAK_SYMMETRY, PLAN_DEGREE_FILTER, PLAN_REFERENCE_ORDER = 1, 2, 4, 8
ROW_SINK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_uint64,
                            ctypes.c_int32)
TEXT_SINK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64)

# every symbol include/gpm_hip.h declares
HIP_SYMBOLS = ["gpm_hip_last_error", "gpm_hip_device_count", "gpm_hip_create", "gpm_hip_destroy",
               "gpm_hip_upload_graph", "gpm_hip_graph_info", "gpm_hip_set_plan", "gpm_hip_set_cursor",
               "gpm_hip_set_roots", "gpm_hip_set_chunks", "gpm_hip_search", "gpm_hip_search_text", "gpm_hip_stats",
               "gpm_hip_kernel_name", "gpm_hip_count_noalloc", "gpm_hip_set_accounting"]

_hip = None
_host = None


def hip_lib():
    """Load libgpm_hip.so (built by `make`); raises when it is missing."""
    global _hip
    if _hip is None:
        path = os.environ.get("GPM_HIP_LIB") or os.path.join(LIB_DIR, "libgpm_hip.so")  # override: kernel experiments
        if not os.path.exists(path):
            raise GpmError(f"{path} is missing: run `make` (or __graft_entry__.build()) first")
        lib = ctypes.CDLL(path)
        lib.gpm_hip_last_error.restype = ctypes.c_char_p
        lib.gpm_hip_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p)]
        lib.gpm_hip_destroy.argtypes = [ctypes.c_void_p]
        lib.gpm_hip_upload_graph.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 5
        lib.gpm_hip_graph_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(gpm_graph_info)]
        lib.gpm_hip_set_plan.argtypes = [ctypes.c_void_p, ctypes.POINTER(gpm_plan)]
        lib.gpm_hip_set_cursor.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.gpm_hip_set_roots.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
        lib.gpm_hip_set_chunks.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        lib.gpm_hip_set_accounting.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.gpm_hip_search.argtypes = [ctypes.c_void_p, ctypes.c_int, ROW_SINK, ctypes.c_void_p,
                                       ctypes.POINTER(ctypes.c_uint64)]
        lib.gpm_hip_search_text.argtypes = [ctypes.c_void_p, TEXT_SINK, ctypes.c_void_p,
                                            ctypes.POINTER(ctypes.c_uint64)]
        lib.gpm_hip_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(gpm_stats)]
        lib.gpm_hip_count_noalloc.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        lib.gpm_hip_kernel_name.restype = ctypes.c_char_p
        lib.gpm_hip_kernel_name.argtypes = [ctypes.c_int]
        _hip = lib
    return _hip


def host_lib():
    """Load libgpm_host.so: parser, CSR builder and planner of the C host."""
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "libgpm_host.so")
        if not os.path.exists(path):
            raise GpmError(f"{path} is missing: run `make` first")
        lib = ctypes.CDLL(path)
        lib.patternFromEdges.restype = ctypes.c_void_p
        lib.patternFromEdges.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
        lib.patternSetLabels.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.freePattern.argtypes = [ctypes.c_void_p]
        lib.freeGraph.argtypes = [ctypes.c_void_p]
        lib.makeGraphOrdering.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.buildPlan.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(gpm_plan), ctypes.c_char_p,
                                  ctypes.c_int]
        lib.graphStatsNew.restype = ctypes.c_void_p
        lib.graphStatsNew.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        lib.graphStatsFree.argtypes = [ctypes.c_void_p]
        lib.buildPlanForGraph.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.POINTER(gpm_plan), ctypes.c_char_p, ctypes.c_int]
        lib.readInput.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                  ctypes.c_char_p, ctypes.c_int]
        lib.free.argtypes = [ctypes.c_void_p]
        lib.writeDataSection.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        _host = lib
    return _host


class HostGraph(ctypes.Structure):
    """Mirror of the C host's Graph (mpi_amd/host/graph.h)."""
    _fields_ = [("nodesCount", ctypes.c_int32), ("edgesCount", ctypes.c_int64),
                ("outOffsets", ctypes.POINTER(ctypes.c_int64)), ("outEdges", ctypes.POINTER(ctypes.c_int32)),
                ("inOffsets", ctypes.POINTER(ctypes.c_int64)), ("inEdges", ctypes.POINTER(ctypes.c_int32)),
                ("labels", ctypes.POINTER(ctypes.c_uint8)), ("symmetric", ctypes.c_int), ("simple", ctypes.c_int)]


def readGraph(path: str):
    """Parse an input file with the C host parser.

    Returns (data, pattern_ptr): data is a dict of numpy CSR arrays
    (0-based ids), pattern_ptr an opaque Pattern* to pass to make_plan /
    free with freePattern.
    """
    lib = host_lib()
    g, p = ctypes.c_void_p(), ctypes.c_void_p()
    err = ctypes.create_string_buffer(256)
    if lib.readInput(path.encode(), ctypes.byref(g), ctypes.byref(p), err, 256) != 0:
        raise GpmError(err.value.decode())
    hg = ctypes.cast(g, ctypes.POINTER(HostGraph)).contents
    n, m = hg.nodesCount, hg.edgesCount
    data = {
        "n": n,
        "out_off": np.ctypeslib.as_array(hg.outOffsets, (n + 1,)).copy(),
        "out_adj": np.ctypeslib.as_array(hg.outEdges, (max(m, 1),))[:m].copy(),
        "in_off": np.ctypeslib.as_array(hg.inOffsets, (n + 1,)).copy(),
        "in_adj": np.ctypeslib.as_array(hg.inEdges, (max(m, 1),))[:m].copy(),
        "labels": np.ctypeslib.as_array(hg.labels, (n,)).copy() if hg.labels else None,
        "symmetric": bool(hg.symmetric), "simple": bool(hg.simple),
    }
    lib.freeGraph(g)
    return data, p


def write_input_csr(path: str, n: int, out_off, out_adj, pattern, data_labels=None, pattern_labels=None):
    """Write an input file from a CSR (0-based ids) with the C host's writer; same
    text as gen.write_input, fast enough for graphs with 10^8+ arcs."""
    from . import gen
    oo = np.ascontiguousarray(out_off, dtype=np.int64)
    oa = np.ascontiguousarray(out_adj, dtype=np.int32)
    if host_lib().writeDataSection(path.encode(), n, oo.ctypes.data, oa.ctypes.data) != 0:
        raise GpmError(f"cannot write {path}")
    P, arcs = pattern
    with open(path, "a") as f:
        f.write(gen._section(P, np.array([a for a, _ in arcs], dtype=np.int64),
                             np.array([b for _, b in arcs], dtype=np.int64)))
        if data_labels is not None:
            f.write("\nlabels\n")
            f.write("\n".join(map(str, np.asarray(data_labels).astype(int).tolist())))
            f.write("\n")
            f.write("\n".join(map(str, list(pattern_labels))))
            f.write("\n")


def make_pattern(pattern, labels=None):
    """(P, [(a, b), ...]) with 1-based arcs -> opaque Pattern* of the C host."""
    lib = host_lib()
    P, arcs = pattern
    s = np.ascontiguousarray(np.array([a - 1 for a, _ in arcs], dtype=np.int32))
    d = np.ascontiguousarray(np.array([b - 1 for _, b in arcs], dtype=np.int32))
    p = lib.patternFromEdges(P, s.ctypes.data, d.ctypes.data, len(arcs))
    if not p:
        raise GpmError("bad pattern")
    if labels is not None:
        l = np.ascontiguousarray(np.asarray(labels, dtype=np.int32))
        lib.patternSetLabels(p, l.ctypes.data)
    return ctypes.c_void_p(p)


def makeGraphOrdering(pattern_ptr):
    """The reference's plan arrays (1-based, index 0 unused): (ordering, parents, ok)."""
    lib = host_lib()
    o = np.zeros(MAXP + 1, dtype=np.int32)
    par = np.zeros(MAXP + 1, dtype=np.int32)
    rc = lib.makeGraphOrdering(pattern_ptr, o.ctypes.data, par.ctypes.data)
    return o, par, rc == 0


class GraphStats:
    """Degree distribution of a data graph for the planner's cost model (GraphStats of mpi_amd/host/plan.h)."""

    def __init__(self, n, out_off, in_off=None):
        oo = np.ascontiguousarray(out_off, dtype=np.int64)
        io = None if in_off is None else np.ascontiguousarray(in_off, dtype=np.int64)
        self.ptr = ctypes.c_void_p(host_lib().graphStatsNew(n, oo.ctypes.data, None if io is None else io.ctypes.data))

    def __del__(self):
        try:
            if self.ptr:
                host_lib().graphStatsFree(self.ptr)
                self.ptr = None
        except Exception:
            pass


def graph_stats(n, out_off, in_off=None) -> GraphStats:
    return GraphStats(n, out_off, in_off)


def make_plan(pattern_ptr, flags: int, stats: GraphStats | None = None) -> gpm_plan:
    """Plan a pattern; with the data graph's degree statistics the binding order is chosen by cost."""
    lib = host_lib()
    plan = gpm_plan()
    err = ctypes.create_string_buffer(256)
    if stats is None:
        rc = lib.buildPlan(pattern_ptr, flags, ctypes.byref(plan), err, 256)
    else:
        rc = lib.buildPlanForGraph(pattern_ptr, flags, stats.ptr, ctypes.byref(plan), err, 256)
    if rc != 0:
        raise GpmError(err.value.decode())
    return plan


class Matcher:
    """One search context on `ngpus` GPUs (gpm_hip_create .. gpm_hip_destroy)."""

    def __init__(self, ngpus: int = 1, devices=None):
        self.lib = hip_lib()
        self.ctx = ctypes.c_void_p()
        dev = (ctypes.c_int * ngpus)(*devices) if devices is not None else None
        self._check(self.lib.gpm_hip_create(ngpus, dev, ctypes.byref(self.ctx)))
        self._plan = None
        self._keep = None

    def _check(self, rc):
        if rc != 0:
            raise GpmError(f"gpm_hip error {rc}: {self.lib.gpm_hip_last_error().decode()}")

    def close(self):
        if self.ctx:
            self.lib.gpm_hip_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, n, out_off, out_adj, in_off=None, in_adj=None, labels=None):
        oo = np.ascontiguousarray(out_off, dtype=np.int64)
        oa = np.ascontiguousarray(out_adj, dtype=np.int32)
        io = ia = lb = None
        if in_off is not None:
            io = np.ascontiguousarray(in_off, dtype=np.int64)
            ia = np.ascontiguousarray(in_adj, dtype=np.int32)
        if labels is not None:
            lb = np.ascontiguousarray(labels, dtype=np.uint8)
        ptr = lambda a: a.ctypes.data if a is not None and a.size else (a.ctypes.data if a is not None else None)
        self._check(self.lib.gpm_hip_upload_graph(self.ctx, n, ptr(oo), ptr(oa), ptr(io), ptr(ia), ptr(lb)))
        self._stats = graph_stats(n, oo, io)
        return self.info()

    def info(self) -> gpm_graph_info:
        gi = gpm_graph_info()
        self._check(self.lib.gpm_hip_graph_info(self.ctx, ctypes.byref(gi)))
        return gi

    def set_pattern(self, pattern, labels=None, symbreak=True, degree_filter=True):
        """Plan `pattern` for the uploaded graph with the C host planner."""
        gi = self.info()
        flags = 0
        if gi.symmetric:
            flags |= PLAN_SYMMETRIC_DATA
        if gi.simple and degree_filter:
            flags |= PLAN_DEGREE_FILTER
        if gi.simple and symbreak:
            flags |= PLAN_BREAK_SYMMETRY
        if not gi.simple:  # duplicate arcs: which arcs are walked decides how often a match is reported
            flags |= PLAN_REFERENCE_ORDER
        p = make_pattern(pattern, labels)
        try:
            plan = make_plan(p, flags, getattr(self, "_stats", None))
        finally:
            host_lib().freePattern(p)
        self.set_plan(plan)
        if plan.n_aut > 1:  # the library copied the table
            host_lib().free(ctypes.cast(plan.aut, ctypes.c_void_p))
            plan.aut = None
        return plan

    def set_plan(self, plan: gpm_plan):
        self._check(self.lib.gpm_hip_set_plan(self.ctx, ctypes.byref(plan)))
        self._plan = plan

    def set_roots(self, lo, hi):
        self._check(self.lib.gpm_hip_set_roots(self.ctx, lo, hi))

    def set_chunks(self, n_chunks):
        self._check(self.lib.gpm_hip_set_chunks(self.ctx, n_chunks))

    def set_accounting(self, on: bool):
        self._check(self.lib.gpm_hip_set_accounting(self.ctx, 1 if on else 0))

    def set_cursor(self, address):
        self._check(self.lib.gpm_hip_set_cursor(self.ctx, address))

    def findMatches(self, count_only=True):
        """Run the search.  count_only: returns the number of embeddings;
        otherwise returns an int32 array [count, P] of 1-based ids in
        pattern-vertex order (every automorphic image is a row)."""
        total = ctypes.c_uint64()
        if count_only:
            self._check(self.lib.gpm_hip_search(self.ctx, 1, ROW_SINK(), None, ctypes.byref(total)))
            return total.value
        chunks = []

        def sink(_user, rows, nrows, ncols):
            chunks.append(np.ctypeslib.as_array(rows, (nrows, ncols)).copy())
            return 0

        cb = ROW_SINK(sink)
        self._check(self.lib.gpm_hip_search(self.ctx, 0, cb, None, ctypes.byref(total)))
        P = self._plan.n_levels
        out = np.concatenate(chunks) if chunks else np.zeros((0, P), dtype=np.int32)
        if len(out) != total.value:
            raise GpmError(f"sink saw {len(out)} rows, search reported {total.value}")
        return out

    def countNoAlloc(self) -> int:
        """Count-only search through the entry point that allocates nothing (gpm_hip_count_noalloc)."""
        total = ctypes.c_uint64()
        self._check(self.lib.gpm_hip_count_noalloc(self.ctx, ctypes.byref(total)))
        return total.value

    def findMatchesText(self) -> bytes:
        """Run the search in output mode with the lines formatted on the GPU; returns the
        output file's bytes (line order unspecified)."""
        total = ctypes.c_uint64()
        blocks, lines = [], [0]

        def sink(_user, text, nbytes, nlines):
            blocks.append(ctypes.string_at(text, nbytes))
            lines[0] += nlines
            return 0

        cb = TEXT_SINK(sink)
        self._check(self.lib.gpm_hip_search_text(self.ctx, cb, None, ctypes.byref(total)))
        if lines[0] != total.value:
            raise GpmError(f"sink saw {lines[0]} lines, search reported {total.value}")
        return b"".join(blocks)

    def stats(self) -> gpm_stats:
        st = gpm_stats()
        self._check(self.lib.gpm_hip_stats(self.ctx, ctypes.byref(st)))
        return st
