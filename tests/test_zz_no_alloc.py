"""gpm_hip_count_noalloc -- the count-only search whose working memory is one
fixed device array -- against gpm_hip_search on the same context, through the
C ABI.  Bar: equal counts (bit-exact)."""
import numpy as np
import pytest
import torch

from mpi_amd import binding, gen
from tests import util
from tests.test_gpu_parity import upload

pytestmark = pytest.mark.gpu

CASES = [
    # (graph spec, pattern, labels, pattern labels) -- the second call of each pair runs on a smaller input
    ({"graph": "rmat", "scale": 11, "ef": 8, "seed": 1}, "clique4", None, None),
    ({"graph": "rmat", "scale": 8, "ef": 8, "seed": 1}, "clique4", None, None),
    ({"graph": "rmat", "scale": 10, "ef": 8, "seed": 1}, "cycle4", None, None),
    ({"graph": "rmat", "scale": 8, "ef": 4, "seed": 2}, "house", None, None),
    ({"graph": "rmat", "scale": 10, "ef": 8, "abcd": (0.7, 0.1, 0.1, 0.1), "seed": 1}, "k6_minus_e", None, None),
    ({"graph": "plaw", "n": 512, "avg_deg": 4, "seed": 1}, "tree5", 4, [0, 1, 2, 3, 1]),
    ({"graph": "rand", "n": 40, "p": [20, 100], "seed": 3, "symmetric": False}, "rev4", None, None),
]


def test_counts_equal_the_allocating_search_on_calls_in_a_row():
    """One context, graphs uploaded one after the other (large, then smaller ones): after each upload
    the no-allocation count equals gpm_hip_search's, three times in a row, and device memory in use does not
    change across the calls."""
    m = binding.Matcher(1)
    for spec, pname, nlab, pl in CASES:
        n, s, d = util.build_graph(spec)
        dl = None if nlab is None else gen.vertex_labels(n, nlab, seed=spec["seed"]).numpy().astype(np.int32)
        upload(m, n, s, d, labels=dl)
        for symbreak in (True, False):
            m.set_pattern(gen.PATTERNS[pname], labels=pl, symbreak=symbreak)
            want = m.findMatches(count_only=True)
            # the first launch of a kernel can make the HIP runtime set up memory of its own (code objects,
            # per-queue spill space), so the memory check brackets the calls after the first
            got = [m.countNoAlloc()]
            free0, _ = torch.cuda.mem_get_info(0)
            got += [m.countNoAlloc(), m.countNoAlloc()]
            free1, _ = torch.cuda.mem_get_info(0)
            assert got == [want, want, want], (spec, pname, symbreak, got, want)
            assert free1 == free0, "device memory in use changed across the no-allocation calls"
    m.close()


def test_two_contexts_share_the_array_one_after_the_other():
    a, b = binding.Matcher(1), binding.Matcher(1)
    n1, s1, d1 = util.build_graph({"graph": "rmat", "scale": 10, "ef": 8, "seed": 1})
    n2, s2, d2 = util.build_graph({"graph": "rmat", "scale": 7, "ef": 8, "seed": 5})
    upload(a, n1, s1, d1)
    upload(b, n2, s2, d2)
    a.set_pattern(gen.PATTERNS["diamond"])
    b.set_pattern(gen.PATTERNS["triangle"])
    wa, wb = a.findMatches(count_only=True), b.findMatches(count_only=True)
    assert [a.countNoAlloc(), b.countNoAlloc(), a.countNoAlloc(), b.countNoAlloc()] == [wa, wb, wa, wb]
    a.close()
    b.close()


def test_sizes_beyond_the_maximum_are_rejected(monkeypatch):
    """A star whose hub has 70 000 neighbours: its list does not fit a frontier of the fixed array."""
    deg = 70_000
    hub = np.zeros(deg, dtype=np.int32)
    leaves = np.arange(1, deg + 1, dtype=np.int32)
    s, d = np.concatenate([hub, leaves]), np.concatenate([leaves, hub])
    order = np.lexsort((d, s))
    m = binding.Matcher(1)
    upload(m, deg + 1, s[order], d[order])
    m.set_pattern(gen.PATTERNS["path3"])
    want = m.findMatches(count_only=True)
    assert want == deg * (deg - 1)
    with pytest.raises(binding.GpmError, match="no-allocation search takes at most"):
        m.countNoAlloc()
    m.close()
    monkeypatch.setenv("GPM_NO_RCCL", "1")  # two workers on one GPU: no communicators
    m2 = binding.Matcher(2, devices=[0, 0])
    with pytest.raises(binding.GpmError):
        m2.countNoAlloc()
    m2.close()


def test_a_context_that_shares_its_cursor_is_rejected():
    """The call walks every root itself; on a context whose chunk cursor other processes share the
    per-process counts would no longer add up to the whole, so it refuses (and works again once the
    cursor is private)."""
    import ctypes
    n, s, d = util.build_graph({"graph": "rmat", "scale": 8, "ef": 8, "seed": 1})
    m = binding.Matcher(1)
    upload(m, n, s, d)
    m.set_pattern(gen.PATTERNS["triangle"])
    want = m.findMatches(count_only=True)
    word = ctypes.c_uint64(0)
    m.set_cursor(ctypes.addressof(word))
    with pytest.raises(binding.GpmError, match="shares a chunk cursor"):
        m.countNoAlloc()
    m.set_cursor(None)
    assert m.countNoAlloc() == want
    m.close()
