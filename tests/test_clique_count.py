"""The independent counter of cliques and 4-cycles (oracle/clique_count.c) against the vectors captured from
the compiled reference (octahedron = K6 minus a perfect matching): it is what pins the benchmark's counts at sizes the reference cannot reach, so it is
itself pinned to every triangle / 4-clique / 4-cycle / octahedron vector the reference did produce."""
import json
import os
import subprocess

import numpy as np
import pytest

from mpi_amd import gen
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_build", "clique_count")
K = {"triangle": "3", "clique4": "4", "cycle4": "c4", "k6_minus_pm": "oct"}


def write_csr(path, n, off, adj):
    with open(path, "wb") as f:
        np.array([n, len(adj)], dtype=np.int64).tofile(f)
        np.asarray(off, dtype=np.int64).tofile(f)
        np.asarray(adj, dtype=np.int32).tofile(f)


def count(path, k, threads=4):
    if not os.path.exists(BIN):  # built with the rest of oracle/ (make -C oracle); here on demand
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, stdout=subprocess.DEVNULL)
    out = subprocess.run([BIN, path, str(k), str(threads)], check=True, capture_output=True, text=True).stdout
    return json.loads(out)


def rmat_specs():
    specs = [g for g in util.golden_specs() + util.golden_large_specs()
             if g["graph"] in ("rmat", "er") and g["pattern"] in K and not g.get("labels") and g.get("symmetric", True)]
    return [g for g in specs if g["graph"] != "rmat" or g["scale"] <= 14]


@pytest.mark.parametrize("spec", rmat_specs(), ids=[g["name"] for g in rmat_specs()])
def test_counter_equals_the_reference(spec, tmp_path):
    if spec["graph"] == "rmat":
        off, adj = gen.rmat(spec["scale"], spec["ef"], abcd=tuple(spec.get("abcd", (0.57, 0.19, 0.19, 0.05))), seed=spec["seed"])
        n = 1 << spec["scale"]
    else:
        off, adj = gen.erdos_renyi(spec["n"], spec["m"], seed=spec["seed"])
        n = spec["n"]
    o, a = gen.csr_to_numpy(off, adj)
    path = str(tmp_path / "g.bin")
    write_csr(path, n, o, a)
    got = count(path, K[spec["pattern"]])
    assert got["embeddings"] == spec["count"], spec["name"]
    assert got["vertices"] == n and got["arcs"] == len(a)


def test_counter_on_a_complete_graph(tmp_path):
    n = 70  # more than 64 higher neighbours: rows of two words
    v = np.arange(n)
    a, b = np.repeat(v, n), np.tile(v, n)
    keep = a != b
    a, b = a[keep], b[keep]
    off = np.zeros(n + 1, dtype=np.int64)
    np.add.at(off, a + 1, 1)
    off = np.cumsum(off)
    path = str(tmp_path / "k.bin")
    write_csr(path, n, off, b)
    assert count(path, 3)["embeddings"] == n * (n - 1) * (n - 2)
    assert count(path, 4)["embeddings"] == n * (n - 1) * (n - 2) * (n - 3)
    assert count(path, "c4")["embeddings"] == n * (n - 1) * (n - 2) * (n - 3)  # every injective map of 4 vertices
    assert count(path, "oct")["embeddings"] == n * (n - 1) * (n - 2) * (n - 3) * (n - 4) * (n - 5)


def test_benchmark_counts_come_from_the_independent_counter():
    """bench.py's expected counts of the clique configurations are the ones the independent counter found
    on the same graphs (tests/golden/full_size_counts.json, made by tests/golden/make_full_size_counts.py)."""
    import bench
    with open(os.path.join(ROOT, "tests", "golden", "full_size_counts.json")) as f:
        full = json.load(f)
    assert sorted(full["configs"]) == ["1", "2", "3", "4", "5"]
    for cfg, entry in full["configs"].items():
        assert bench.EXPECTED_FULL[int(cfg)] == entry["embeddings"], cfg
