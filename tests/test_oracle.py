"""The CPU oracle against the compiled reference's golden vectors.

Golden outputs were captured from the reference built from its own sources
and run as `mpirun -np 2` (tests/golden/make_golden.py).  With one worker the
reference's output order is deterministic, so the oracle must reproduce the
files byte for byte, not only as a multiset.
"""
import numpy as np
import pytest

from mpi_amd import gen
from tests import util

SMALL = util.small_cases()
GOLDEN = util.golden_specs()
# the heaviest vectors only run through the binary-search variant of the oracle
LIGHT = [g for g in GOLDEN if g["count"] <= 200000]


@pytest.mark.parametrize("name", SMALL)
@pytest.mark.parametrize("fast", [0, 1])
def test_small_cases_byte_identical(name, fast):
    inp, out = util.small_paths(name)
    with open(inp) as f:
        text = f.read()
    with open(out, "rb") as f:
        want = f.read()
    assert util.oracle_run_text(text, fast) == want


@pytest.mark.parametrize("spec", GOLDEN, ids=[g["name"] for g in GOLDEN])
def test_generated_cases_match_reference_md5(spec):
    text, dl, pl = util.build_input(spec)
    assert util.md5(text.encode()) == spec["input_md5"], "generator drifted from the captured input"
    got = util.oracle_run_text(text, fast=1)
    assert got.count(b"\n") == spec["count"]
    assert util.md5(got) == spec["output_md5"]


@pytest.mark.parametrize("spec", LIGHT, ids=[g["name"] for g in LIGHT])
def test_linear_scan_variant_agrees(spec):
    text, _, _ = util.build_input(spec)
    assert util.md5(util.oracle_run_text(text, fast=0)) == spec["output_md5"]


@pytest.mark.parametrize("pname", ["edge", "tri_dir", "triangle", "path3", "cycle4", "clique4", "k4_dag", "rev4",
                                   "tree5_dir"])
@pytest.mark.parametrize("symmetric", [False, True])
def test_oracle_equals_brute_force(pname, symmetric):
    n = 8
    s, d = util.rand_digraph(n, 40, 100, 11, symmetric)
    cnt, rows, _ = util.oracle_matches(n, s, d, gen.PATTERNS[pname], fast=0)
    want = util.brute_force(n, s, d, gen.PATTERNS[pname])
    assert cnt == len(want)
    assert sorted(map(tuple, rows.tolist())) == want


def test_root_ranges_partition_the_result():
    n, s, d = util.build_graph({"graph": "rmat", "scale": 8, "ef": 8, "seed": 1})
    pat = gen.PATTERNS["clique4"]
    total, rows, _ = util.oracle_matches(n, s, d, pat)
    parts = [util.oracle_matches(n, s, d, pat, root_lo=lo, root_hi=hi)[1]
             for lo, hi in [(1, 100), (101, 101), (102, n)]]
    assert sum(len(p) for p in parts) == total
    assert np.array_equal(np.concatenate(parts), rows)


def test_label_filter_is_a_post_filter():
    spec = {"graph": "plaw", "n": 256, "avg_deg": 6, "seed": 2}
    n, s, d = util.build_graph(spec)
    dl = gen.vertex_labels(n, 4, seed=2).numpy().astype(np.int32)
    pl = [0, 1, 2]
    pat = gen.PATTERNS["path3"]
    _, rows, _ = util.oracle_matches(n, s, d, pat)
    keep = np.all(dl[rows - 1] == np.array(pl)[None, :], axis=1)
    _, lrows, _ = util.oracle_matches(n, s, d, pat, data_labels=dl, pattern_labels=pl)
    assert np.array_equal(rows[keep], lrows)


@pytest.mark.parametrize("pname,plabels", [("path3", [0, 1, 2]), ("tree5", [0, 1, 2, 3, 1]), ("triangle", [1, 1, 0]),
                                           ("tree5_dir", [0, 0, 1, 2, 1]), ("cycle4", [0, 1, 0, 1])])
def test_label_pruning_gives_the_post_filtered_rows_in_the_same_order(pname, plabels):
    """fast = 2 tests labels while vertices are bound; the rows and their order are those of the
    post-filter (fast = 1), which in turn is pinned to the reference's filtered output."""
    n, s, d = util.build_graph({"graph": "plaw", "n": 256, "avg_deg": 6, "seed": 2})
    if pname == "tree5_dir":
        s, d = util.rand_digraph(60, 1, 5, 4, False)
        n = 60
    dl = gen.vertex_labels(n, 4, seed=2).numpy().astype(np.int32)
    pat = gen.PATTERNS[pname]
    c1, r1, _ = util.oracle_matches(n, s, d, pat, fast=1, data_labels=dl, pattern_labels=plabels)
    c2, r2, _ = util.oracle_matches(n, s, d, pat, fast=2, data_labels=dl, pattern_labels=plabels)
    assert c1 == c2 and np.array_equal(r1, r2)


LARGE_SMALL_ENOUGH = [g for g in util.golden_large_specs() if g["count"] <= 5_000_000]


@pytest.mark.parametrize("spec", LARGE_SMALL_ENOUGH, ids=[g["name"] for g in LARGE_SMALL_ENOUGH])
def test_oracle_equals_the_large_vectors_it_can_afford(spec):
    """Count and row checksum of the captured vectors with up to a few million matches (the labelled
    ones with the label test applied while binding; the 16-label vector on the small graph was
    captured from the reference's unlabelled output, filtered)."""
    n, s, d = util.build_graph(spec)
    dl, pl = util.build_labels(spec, n)
    cnt, rows, _ = util.oracle_matches(n, s, d, gen.PATTERNS[spec["pattern"]], fast=2 if dl is not None else 1,
                                       data_labels=dl, pattern_labels=pl, cap=spec["count"] + 1)
    assert cnt == spec["count"]
    assert util.rows_multiset_hash(rows) == spec["rows_hash"]
