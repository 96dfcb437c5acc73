"""Deterministic synthetic graph generators and the text-format writer.

Every generator is a pure function of (parameters, seed) built on a
counter-based splitmix64 hash, evaluated with integer torch ops only, so the
same graph comes out on CPU and on a GPU.  Graphs are simple (no self loops,
no duplicate edges) and symmetrised (u->v and v->u both present), ids are
0-based here and become 1-based in the text format.

The text format is the reference's input format (reference: readGraph,
src/graph.c:38-67; preprocessGraph, src/gpm-par.c:267-286): vertex count,
then one block per vertex that has out-edges ("<v> <deg>" and deg neighbour
lines), a blank line, then the pattern in the same grammar.  Vertices without
out-edges are not listed (a "<v> 0" block crashes the reference).
"""
from __future__ import annotations

import io
from typing import Iterable, Sequence

import numpy as np
import torch

_MASK63 = (1 << 63) - 1


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    """Logical shift right of int64 bit patterns."""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _i64(v: int) -> int:
    """Python int -> two's-complement int64 value."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


_GOLD = _i64(0x9E3779B97F4A7C15)
_M1 = _i64(0xBF58476D1CE4E5B9)
_M2 = _i64(0x94D049BB133111EB)


def splitmix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 bit patterns (wrap-around arithmetic)."""
    z = x + _GOLD
    z = (z ^ _lsr(z, 30)) * _M1
    z = (z ^ _lsr(z, 27)) * _M2
    return z ^ _lsr(z, 31)


def _rand31(counter: torch.Tensor, seed: int) -> torch.Tensor:
    """31-bit uniform integers from a counter and a seed."""
    return _lsr(splitmix64(counter ^ _i64(seed * 0xD6E8FEB86659FD93)), 33)


def _scramble(v: torch.Tensor, bits: int, seed: int) -> torch.Tensor:
    """Bijection on [0, 2^bits): hides the id/degree correlation of R-MAT."""
    mask = (1 << bits) - 1
    a = (splitmix64(torch.tensor([seed + 17], dtype=torch.int64)).item() | 1) & mask
    b = splitmix64(torch.tensor([seed + 91], dtype=torch.int64)).item() & mask
    v = (v * a + b) & mask
    v = v ^ (v >> max(1, bits // 2))
    v = (v * ((a * 5 + 2) | 1)) & mask
    return v


def _finish(u: torch.Tensor, v: torch.Tensor, n: int, directed: bool = False, multi: bool = False):
    """Drop self loops, symmetrise, dedupe -> (offsets int64[n+1], adj int32[m]).
    directed: keep the sampled arcs u -> v as they are (no reverse arcs); multi: keep duplicate arcs."""
    keep = u != v
    u, v = u[keep], v[keep]
    key = u * n + v if directed else torch.cat([u * n + v, v * n + u])
    del u, v
    key = torch.sort(key).values if multi else torch.unique(key)  # ascending
    src = torch.div(key, n, rounding_mode="floor")
    adj = (key - src * n).to(torch.int32)
    del key
    deg = torch.bincount(src, minlength=n)
    off = torch.zeros(n + 1, dtype=torch.int64, device=adj.device)
    off[1:] = torch.cumsum(deg, 0)
    return off, adj


def rmat(scale: int, edge_factor: int, abcd=(0.57, 0.19, 0.19, 0.05), seed: int = 1,
         device: str | torch.device = "cpu", chunk: int = 1 << 25, directed: bool = False, multi: bool = False):
    """R-MAT graph: 2^scale vertices, edge_factor * 2^scale sampled edges -- undirected (both arcs listed) unless
    `directed`, which keeps each sample as the one arc u -> v; `multi` keeps arcs sampled more than once."""
    n = 1 << scale
    m = edge_factor * n
    ta = int(abcd[0] * (1 << 31))
    tb = ta + int(abcd[1] * (1 << 31))
    tc = tb + int(abcd[2] * (1 << 31))
    us, vs = [], []
    for e0 in range(0, m, chunk):
        e = torch.arange(e0, min(m, e0 + chunk), dtype=torch.int64, device=device)
        u = torch.zeros_like(e)
        v = torch.zeros_like(e)
        for lvl in range(scale):
            r = _rand31(e * 64 + lvl, seed)
            ubit = (r >= tb).to(torch.int64)           # quadrants c, d: lower half of rows
            vbit = ((r >= ta) & (r < tb) | (r >= tc)).to(torch.int64)  # quadrants b, d
            u = (u << 1) | ubit
            v = (v << 1) | vbit
        us.append(_scramble(u, scale, seed))
        vs.append(_scramble(v, scale, seed))
    return _finish(torch.cat(us), torch.cat(vs), n, directed, multi)


def erdos_renyi(n: int, m: int, seed: int = 1, device: str | torch.device = "cpu"):
    """G(n, m): exactly m distinct undirected edges (first m distinct samples)."""
    over = int(m * 1.2) + 64
    e = torch.arange(over, dtype=torch.int64, device=device)
    u = _rand31(e * 2, seed) % n
    v = _rand31(e * 2 + 1, seed) % n
    lo, hi = torch.minimum(u, v), torch.maximum(u, v)
    key = (lo * n + hi)[lo != hi].cpu().numpy()
    _, first = np.unique(key, return_index=True)
    first.sort()
    if len(first) < m:
        raise ValueError("oversampling too small")
    key = torch.from_numpy(key[first[:m]]).to(device)
    lo = torch.div(key, n, rounding_mode="floor")
    return _finish(lo, key - lo * n, n)


def power_law(n: int, avg_deg: int = 16, exponent: float = 2.1, seed: int = 1,
              device: str | torch.device = "cpu", chunk: int = 1 << 25):
    """Chung-Lu style graph: endpoint i drawn with weight (i+1)^(-1/(exponent-1))."""
    m = n * avg_deg // 2
    w = torch.arange(1, n + 1, dtype=torch.float64, device=device).pow(-1.0 / (exponent - 1.0))
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    us, vs = [], []
    for e0 in range(0, m, chunk):
        e = torch.arange(e0, min(m, e0 + chunk), dtype=torch.int64, device=device)
        ru = _rand31(e * 2, seed).to(torch.float64) / float(1 << 31)
        rv = _rand31(e * 2 + 1, seed).to(torch.float64) / float(1 << 31)
        us.append(torch.searchsorted(cdf, ru).clamp_(max=n - 1))
        vs.append(torch.searchsorted(cdf, rv).clamp_(max=n - 1))
    bits = max(1, (n - 1).bit_length())
    u, v = torch.cat(us), torch.cat(vs)
    if n == 1 << bits:
        u, v = _scramble(u, bits, seed), _scramble(v, bits, seed)
    return _finish(u, v, n)


def vertex_labels(n: int, n_labels: int = 16, seed: int = 1, device: str | torch.device = "cpu"):
    """Uniform labels in [0, n_labels) for vertices 0..n-1 (uint8)."""
    v = torch.arange(n, dtype=torch.int64, device=device)
    return (_rand31(v, seed + 1000) % n_labels).to(torch.uint8)


# ---- patterns (1-based directed arc lists, in listing order) ---------------

def _both(pairs: Iterable[Sequence[int]]):
    arcs = []
    for a, b in pairs:
        arcs.append((a, b))
        arcs.append((b, a))
    return sorted(set(arcs))


PATTERNS = {
    "edge": (2, [(1, 2)]),
    "edge_und": (2, _both([(1, 2)])),
    "tri_dir": (3, [(1, 2), (2, 3), (3, 1)]),
    "triangle": (3, _both([(1, 2), (2, 3), (1, 3)])),
    "path3": (3, _both([(1, 2), (2, 3)])),
    "star4": (4, _both([(1, 2), (1, 3), (1, 4)])),
    "path4": (4, _both([(1, 2), (2, 3), (3, 4)])),
    "cycle4": (4, _both([(1, 2), (2, 3), (3, 4), (4, 1)])),
    "clique4": (4, _both([(a, b) for a in range(1, 5) for b in range(a + 1, 5)])),
    "diamond": (4, _both([(1, 2), (1, 3), (2, 3), (2, 4), (3, 4)])),
    "k4_dag": (4, [(a, b) for a in range(1, 5) for b in range(a + 1, 5)]),
    "rev4": (4, [(2, 1), (3, 1), (3, 2), (4, 3)]),
    "tree5": (5, _both([(1, 2), (1, 3), (3, 4), (3, 5)])),
    "tree5_dir": (5, [(1, 2), (1, 3), (3, 4), (5, 3)]),
    "clique5": (5, _both([(a, b) for a in range(1, 6) for b in range(a + 1, 6)])),
    "clique6": (6, _both([(a, b) for a in range(1, 7) for b in range(a + 1, 7)])),
    "house": (5, _both([(1, 2), (2, 3), (3, 4), (4, 1), (1, 5), (2, 5)])),
    # K6 minus a perfect matching (the octahedron): 12 undirected edges
    "k6_minus_pm": (6, _both([(a, b) for a in range(1, 7) for b in range(a + 1, 7)
                              if (a, b) not in ((1, 2), (3, 4), (5, 6))])),
    "k6_minus_e": (6, _both([(a, b) for a in range(1, 7) for b in range(a + 1, 7) if (a, b) != (5, 6)])),
}


def csr_to_numpy(off: torch.Tensor, adj: torch.Tensor):
    return off.cpu().numpy().astype(np.int64), adj.cpu().numpy().astype(np.int32)


def edges_from_csr(off: np.ndarray, adj: np.ndarray):
    """(src, dst) int32 arrays, 0-based, in CSR order."""
    deg = np.diff(off)
    src = np.repeat(np.arange(len(deg), dtype=np.int32), deg)
    return src, adj.astype(np.int32)


def _section(n: int, src: np.ndarray, dst: np.ndarray) -> str:
    """One section of the text format from 1-based arcs in listing order."""
    out = io.StringIO()
    out.write(f"{n}\n")
    if len(src):
        order = np.argsort(src, kind="stable")
        s, d = np.asarray(src)[order], np.asarray(dst)[order]
        starts = np.flatnonzero(np.r_[True, s[1:] != s[:-1]])
        ends = np.r_[starts[1:], len(s)]
        for a, b in zip(starts, ends):
            out.write(f"{s[a]} {b - a}\n")
            out.write("\n".join(map(str, d[a:b].tolist())))
            out.write("\n")
    return out.getvalue()


def write_input(path: str, n: int, src0: np.ndarray, dst0: np.ndarray, pattern,
                data_labels=None, pattern_labels=None) -> None:
    """Write a reference-format input file.

    src0/dst0: 0-based data arcs; pattern: (P, [(a, b), ...]) with 1-based arcs.
    Labels (optional) go into a trailer after the pattern's blank line, which
    the reference ignores: the word "labels", then n + P integers, one per line.
    """
    P, arcs = pattern
    ps = np.array([a for a, _ in arcs], dtype=np.int64)
    pd = np.array([b for _, b in arcs], dtype=np.int64)
    with open(path, "w") as f:
        f.write(_section(n, np.asarray(src0, dtype=np.int64) + 1, np.asarray(dst0, dtype=np.int64) + 1))
        f.write("\n")
        f.write(_section(P, ps, pd))
        if data_labels is not None:
            f.write("\nlabels\n")
            f.write("\n".join(map(str, np.asarray(data_labels).astype(int).tolist())))
            f.write("\n")
            f.write("\n".join(map(str, list(pattern_labels))))
            f.write("\n")
