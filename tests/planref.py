"""Pure-Python interpreter of a gpm_plan, for checking the C planner's output
on tiny graphs without a GPU (test infrastructure)."""
import numpy as np


def csr_lists(n, src0, dst0):
    """Sorted out/in adjacency lists (duplicates kept), 0-based."""
    out = [[] for _ in range(n)]
    inn = [[] for _ in range(n)]
    for a, b in zip(np.asarray(src0).tolist(), np.asarray(dst0).tolist()):
        out[a].append(b)
        inn[b].append(a)
    return [sorted(x) for x in out], [sorted(x) for x in inn]


def run_plan(plan, n, out, inn, labels=None, symmetric=False):
    """Enumerate the plan's canonical matches; returns rows in pattern-vertex
    order (0-based data ids), one per canonical match, before automorphism
    expansion.  Vertex order for the bounds is the plain id order."""
    P = plan.n_levels
    lists = (out, out) if symmetric else (out, inn)
    sets = tuple([set(x) for x in l] for l in lists)
    res = []
    m = [0] * P

    def ok_vertex(k, c):
        if plan.label[k] >= 0 and labels[c] != plan.label[k]:
            return False
        if len(set(out[c]) - {c}) < plan.min_out_deg[k]:
            return False
        if len(set(inn[c]) - {c}) < plan.min_in_deg[k]:
            return False
        return True

    def rec(k):
        if k == P:
            row = [0] * P
            for t in range(P):
                row[plan.order[t]] = m[t]
            res.append(row)
            return
        for c in lists[plan.src_dir[k]][m[plan.src_level[k]]]:  # with multiplicity
            if c in m[:k] or not ok_vertex(k, c):
                continue
            if any(c not in sets[plan.cons_dir[k][j]][m[plan.cons_level[k][j]]] for j in range(plan.n_cons[k])):
                continue
            if any(c <= m[plan.lower_level[k][j]] for j in range(plan.n_lower[k])):
                continue
            if any(c >= m[plan.upper_level[k][j]] for j in range(plan.n_upper[k])):
                continue
            m[k] = c
            rec(k + 1)

    for v in range(n):
        if ok_vertex(0, v):
            m[0] = v
            rec(1)
    return res


def expand_automorphisms(plan, rows):
    P = plan.n_levels
    if plan.n_aut <= 1:
        return [tuple(r) for r in rows]
    out = []
    for r in rows:
        for i in range(plan.n_aut):
            out.append(tuple(r[plan.aut[i * P + v]] for v in range(P)))
    return out
