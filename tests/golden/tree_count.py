"""Independent closed-form count of the benchmark's labelled 5-vertex tree (pattern "tree5": a centre 1 with a
leaf 2 and a second centre 3 that has the leaves 4 and 5) in a simple symmetric graph, with numpy on the CPU.

TEST INFRASTRUCTURE: it pins the count bench.py expects of configuration 4 at full size, where neither the
reference (no labels, and 10^12 unlabelled matches to filter) nor the enumerating oracle can go; it shares
nothing with the HIP library's leaf-suffix kernel but the definition of a match (reference: checkNodeMatches,
src/gpm-par.c:88-112: injective, every pattern arc present), and tests/test_tree_count.py pins it to the
labelled and unlabelled vectors the reference and the oracle did produce.

For an arc (u1, u3) with the labels of pattern vertices 1 and 3, let X2 = the neighbours of u1 with label l2
other than u3, X4 / X5 = the neighbours of u3 with label l4 / l5 other than u1.  The matches over that arc are
the triples (m2, m4, m5) of X2 x X4 x X5 that are pairwise distinct:
|X2||X4||X5| - |X2^X4||X5| - |X2^X5||X4| - |X4^X5||X2| + 2|X2^X4^X5|, where X4^X5 is X4 when l4 = l5 and empty
otherwise, and X2^X4 (X2^X5) is the set of common neighbours of u1 and u3 with label l2 when l2 = l4 (l5).
Unlabelled: every label equal."""
import numpy as np


def count_tree5(n, off, adj, labels=None, pattern_labels=None):
    """Embeddings of tree5 (python int).  off/adj: CSR of a simple symmetric graph, lists ascending."""
    off = np.asarray(off, dtype=np.int64)
    adj = np.asarray(adj, dtype=np.int64)
    lab = np.zeros(n, dtype=np.int64) if labels is None else np.asarray(labels, dtype=np.int64)
    l1, l2, l3, l4, l5 = (0, 0, 0, 0, 0) if pattern_labels is None else [int(x) for x in pattern_labels]
    src = np.repeat(np.arange(n, dtype=np.int64), np.diff(off))

    def neighbours_with(label):
        return np.bincount(src[lab[adj] == label], minlength=n)

    cnt = {l: neighbours_with(l) for l in {l2, l4, l5}}
    sel = (lab[src] == l1) & (lab[adj] == l3)
    u1s, u3s = src[sel], adj[sel]
    need_common = l2 == l4 or l2 == l5
    if need_common:  # lists restricted to label l2, for the common-neighbour counts
        keep = lab[adj] == l2
        f_adj = adj[keep]
        f_off = np.zeros(n + 1, dtype=np.int64)
        f_off[1:] = np.cumsum(np.bincount(src[keep], minlength=n))
    total = 0
    for u1, u3 in zip(u1s.tolist(), u3s.tolist()):
        a2 = int(cnt[l2][u1]) - (1 if lab[u3] == l2 else 0)
        a4 = int(cnt[l4][u3]) - (1 if lab[u1] == l4 else 0)
        a5 = int(cnt[l5][u3]) - (1 if lab[u1] == l5 else 0)
        if a2 <= 0 or a4 <= 0 or a5 <= 0:
            continue
        t = 0
        if need_common:
            t = len(np.intersect1d(f_adj[f_off[u1]:f_off[u1 + 1]], f_adj[f_off[u3]:f_off[u3 + 1]], assume_unique=True))
        c24 = t if l2 == l4 else 0
        c25 = t if l2 == l5 else 0
        c45 = a4 if l4 == l5 else 0
        c245 = t if l2 == l4 == l5 else 0
        total += a2 * a4 * a5 - c24 * a5 - c25 * a4 - c45 * a2 + 2 * c245
    return total
