"""Shared test helpers: fixture construction and the ctypes binding of the
CPU oracle (test infrastructure; the product never imports this)."""
from __future__ import annotations

import ctypes
import hashlib
import io
import json
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

from mpi_amd import gen  # noqa: E402


def md5(b: bytes) -> str:
    return hashlib.md5(b).hexdigest()


def golden_specs():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


def golden_large_specs():
    """Vectors whose outputs are too large to keep: count and order-independent row checksum
    (tests/golden/make_golden_large.py)."""
    with open(os.path.join(GOLDEN_DIR, "golden_large.json")) as f:
        return json.load(f)


def small_cases():
    d = os.path.join(GOLDEN_DIR, "small")
    return sorted(f[:-3] for f in os.listdir(d) if f.endswith(".in"))


def small_paths(name):
    d = os.path.join(GOLDEN_DIR, "small")
    return os.path.join(d, name + ".in"), os.path.join(d, name + ".out")


def rand_digraph(n, p_num, p_den, seed, symmetric):
    import torch
    idx = torch.arange(n * n, dtype=torch.int64)
    r = gen._rand31(idx, seed) % p_den
    u = torch.div(idx, n, rounding_mode="floor")
    v = idx - u * n
    keep = (r < p_num) & (u != v)
    u, v = u[keep].numpy(), v[keep].numpy()
    if symmetric:
        key = np.unique(np.r_[u * n + v, v * n + u])
        u, v = key // n, key % n
    return u.astype(np.int32), v.astype(np.int32)


def build_graph(spec):
    """spec -> (n, src0, dst0) with 0-based arcs in listing order."""
    kind = spec["graph"]
    if kind == "rand":
        s, d = rand_digraph(spec["n"], spec["p"][0], spec["p"][1], spec["seed"], spec["symmetric"])
        return spec["n"], s, d
    if kind == "er":
        off, adj = gen.erdos_renyi(spec["n"], spec["m"], seed=spec["seed"])
        n = spec["n"]
    elif kind == "rmat":
        off, adj = gen.rmat(spec["scale"], spec["ef"], abcd=tuple(spec.get("abcd", (0.57, 0.19, 0.19, 0.05))),
                            seed=spec["seed"], directed=bool(spec.get("directed")), multi=bool(spec.get("multi")))
        n = 1 << spec["scale"]
    elif kind == "plaw":
        off, adj = gen.power_law(spec["n"], avg_deg=spec["avg_deg"], seed=spec["seed"])
        n = spec["n"]
    else:
        raise ValueError(kind)
    o, a = gen.csr_to_numpy(off, adj)
    s, d = gen.edges_from_csr(o, a)
    return n, s, d


def build_labels(spec, n):
    if "labels" not in spec:
        return None, None
    dl = gen.vertex_labels(n, spec["labels"], seed=spec["seed"]).numpy().astype(np.int32)
    return dl, list(spec["pattern_labels"])


def build_input(spec):
    """spec -> (input text, data labels or None, pattern labels or None)."""
    n, s, d = build_graph(spec)
    dl, pl = build_labels(spec, n)
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "in.txt")
        gen.write_input(p, n, s, d, gen.PATTERNS[spec["pattern"]], dl, pl)
        with open(p) as f:
            return f.read(), dl, pl


# ---- oracle binding ---------------------------------------------------------

class OStats(ctypes.Structure):
    _fields_ = [("matches", ctypes.c_longlong),
                ("extended", ctypes.c_longlong * 12),
                ("alg_bytes", ctypes.c_longlong)]


_ORACLE = None


def oracle_lib():
    """Load (building if needed) oracle/_build/liboracle.so."""
    global _ORACLE
    if _ORACLE is None:
        so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True,
                           stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        lib.oracle_graph_from_edges.restype = ctypes.c_void_p
        lib.oracle_graph_from_edges.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
        lib.oracle_graph_free.argtypes = [ctypes.c_void_p]
        lib.oracle_make_plan.argtypes = [ctypes.c_void_p]
        lib.oracle_search.restype = ctypes.c_longlong
        lib.oracle_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_longlong, ctypes.c_void_p, ctypes.POINTER(OStats)]
        lib.oracle_run_file.restype = ctypes.c_longlong
        lib.oracle_run_file.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        _ORACLE = lib
    return _ORACLE


def oracle_run_text(text: str, fast: int = 0) -> bytes:
    """Run the oracle's whole-program path on an input text; returns the output file bytes."""
    lib = oracle_lib()
    with tempfile.TemporaryDirectory() as tmp:
        i, o = os.path.join(tmp, "in.txt"), os.path.join(tmp, "out.txt")
        with open(i, "w") as f:
            f.write(text)
        r = lib.oracle_run_file(i.encode(), o.encode(), fast)
        if r < 0:
            raise RuntimeError("oracle failed")
        with open(o, "rb") as f:
            return f.read()


def oracle_matches(n, src0, dst0, pattern, fast=1, data_labels=None, pattern_labels=None,
                   root_lo=1, root_hi=None, want_rows=True, cap=None):
    """Run the oracle on 0-based arcs; returns (count, rows ndarray [count, P] 1-based, stats)."""
    lib = oracle_lib()
    P, arcs = pattern
    s1 = np.ascontiguousarray(np.asarray(src0, dtype=np.int32) + 1)
    d1 = np.ascontiguousarray(np.asarray(dst0, dtype=np.int32) + 1)
    g = lib.oracle_graph_from_edges(n, s1.ctypes.data, d1.ctypes.data, len(s1))
    ps = np.array([a for a, _ in arcs], dtype=np.int32)
    pd = np.array([b for _, b in arcs], dtype=np.int32)
    p = lib.oracle_graph_from_edges(P, ps.ctypes.data, pd.ctypes.data, len(ps))
    assert g and p
    if lib.oracle_make_plan(p) != 0:
        raise ValueError("pattern not connected from vertex 1")
    dl = pl = None
    if data_labels is not None:
        dl = np.ascontiguousarray(np.r_[0, np.asarray(data_labels)].astype(np.int32))
        pl = np.ascontiguousarray(np.r_[0, np.asarray(pattern_labels)].astype(np.int32))
    st = OStats()
    rows = None
    if want_rows:
        if cap is None:
            cap = lib.oracle_search(g, p, fast, dl.ctypes.data if dl is not None else None,
                                    pl.ctypes.data if pl is not None else None, root_lo,
                                    root_hi if root_hi is not None else n, None, 0, None, ctypes.byref(st))
        rows = np.zeros((max(cap, 1), P), dtype=np.int32)
    cnt = lib.oracle_search(g, p, fast, dl.ctypes.data if dl is not None else None,
                            pl.ctypes.data if pl is not None else None, root_lo,
                            root_hi if root_hi is not None else n,
                            rows.ctypes.data if rows is not None else None,
                            len(rows) if rows is not None else 0, None, ctypes.byref(st))
    lib.oracle_graph_free(g)
    lib.oracle_graph_free(p)
    if rows is not None:
        rows = rows[:min(cnt, len(rows))]
    return cnt, rows, st


def rows_to_text(rows: np.ndarray) -> bytes:
    out = io.BytesIO()
    np.savetxt(out, rows, fmt="%d", delimiter=" ")
    return out.getvalue()


def sorted_lines_md5(b: bytes) -> str:
    return md5(b"".join(sorted(b.splitlines(keepends=True))))


_SM_GOLD, _SM_M1, _SM_M2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)


def _mix64(z):
    z = z + _SM_GOLD
    z = (z ^ (z >> np.uint64(30))) * _SM_M1
    z = (z ^ (z >> np.uint64(27))) * _SM_M2
    return z ^ (z >> np.uint64(31))


def rows_multiset_hash(rows: np.ndarray) -> int:
    """Order-independent checksum of match rows: every row is hashed (a chain of 64-bit mixes over
    its columns), the row hashes are added modulo 2^64.  Equal multisets of rows give equal sums
    whatever the order the rows arrive in, so it can be accumulated block by block."""
    with np.errstate(over="ignore"):
        r = np.ascontiguousarray(rows).astype(np.uint64)
        h = np.zeros(len(r), dtype=np.uint64)
        for c in range(r.shape[1]):
            h = _mix64(h ^ r[:, c])
        return int(h.sum(dtype=np.uint64))


def text_multiset_hash(stream, ncols: int, block: int = 1 << 26):
    """(line count, rows_multiset_hash) of output lines read from a binary stream, block by block."""
    count, total, carry = 0, 0, b""
    while True:
        buf = stream.read(block)
        if not buf:
            break
        buf = carry + buf
        cut = buf.rfind(b"\n") + 1
        carry, buf = buf[cut:], buf[:cut]
        if not buf:
            continue
        vals = np.fromstring(buf.decode(), dtype=np.int64, sep=" ")
        rows = vals.reshape(-1, ncols)
        count += len(rows)
        total = (total + rows_multiset_hash(rows)) & ((1 << 64) - 1)
    assert not carry.strip(), "output does not end with a newline"
    return count, total


def brute_force(n, src0, dst0, pattern):
    """All injective maps V(P)->V(G) preserving every pattern arc (self loops
    ignored), by exhaustive search; tiny graphs only.  Duplicate data arcs are
    collapsed, so use it on simple graphs."""
    import itertools
    P, arcs = pattern
    es = set(zip((np.asarray(src0) + 1).tolist(), (np.asarray(dst0) + 1).tolist()))
    arcs = [(a, b) for a, b in arcs if a != b]
    res = []
    for perm in itertools.permutations(range(1, n + 1), P):
        if all((perm[a - 1], perm[b - 1]) in es for a, b in arcs):
            res.append(perm)
    return sorted(res)
