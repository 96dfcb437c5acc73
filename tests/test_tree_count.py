"""The independent closed-form count of the labelled 5-vertex tree (tests/golden/tree_count.py) against the
vectors of the compiled reference and of the pinned oracle: it pins configuration 4's count at full size."""
import json
import os

import numpy as np
import pytest

from mpi_amd import gen
from tests import util
from tests.golden.tree_count import count_tree5

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPECS = [g for g in util.golden_specs() + util.golden_large_specs()
         if g["pattern"] == "tree5" and g.get("symmetric", True) is not False]


@pytest.mark.parametrize("spec", SPECS, ids=[g["name"] for g in SPECS])
def test_tree_count_equals_the_vectors(spec):
    n, s, d = util.build_graph(spec)
    dl, pl = util.build_labels(spec, n)
    order = np.lexsort((d, s))
    s, d = np.asarray(s)[order], np.asarray(d)[order]
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum(np.bincount(s, minlength=n))
    assert count_tree5(n, off, d, dl, pl) == spec["count"], spec["name"]


def test_benchmark_count_of_the_labelled_tree_comes_from_the_closed_form():
    import bench
    with open(os.path.join(ROOT, "tests", "golden", "full_size_counts.json")) as f:
        full = json.load(f)
    assert bench.EXPECTED_FULL[4] == full["configs"]["4"]["embeddings"]
