"""The benchmark configurations at FULL size on the GPU, count-only, against counts made outside the
library (tests/golden/full_size_counts.json: an independent CPU counter for triangles, 4-cycles, 4-cliques
and octahedra, a closed form for the labelled tree; both pinned to the reference's vectors by
tests/test_clique_count.py and tests/test_tree_count.py).

The reference enumerates every root itself (reference: src/gpm-par.c:245-252) and cannot finish these sizes;
the counts here are what its output files would hold lines of.  Graphs are generated on the GPU by the
benchmark's own generator, so a case costs seconds.  The headline configuration (R-MAT scale 24, 4-clique) is
bench.py's default run, which checks its count itself."""
import json
import os
import subprocess

import pytest

import bench
from mpi_amd import binding, gen
from tests import util

pytestmark = pytest.mark.gpu

with open(os.path.join(util.GOLDEN_DIR, "full_size_counts.json")) as _f:
    FULL = json.load(_f)["configs"]
KIND_NAMES = {1: "k_clique", 2: "k_cycle4", 4: "k_tail", 5: "k_octa"}


def _generate(config):
    import torch
    cfg = bench.CONFIGS[config]
    dev = torch.device("cuda", 0)
    n, off, adj = bench.generate(cfg["graph"], dev)
    labels = None
    if "labels" in cfg:
        labels = gen.vertex_labels(n, cfg["labels"], seed=cfg["graph"]["seed"], device=dev).cpu().numpy()
    off_h, adj_h = off.cpu().numpy(), adj.cpu().numpy()
    del off, adj
    torch.cuda.empty_cache()
    return cfg, n, off_h, adj_h, labels


@pytest.mark.parametrize("config", [1, 2, 4, 5])
def test_configuration_at_full_size(config, monkeypatch):
    """Configurations 1, 2, 4 and 5 as BASELINE.json states them: the count of the search, the kernel family
    that did the work, and the vertex and arc numbers of the generated graph, all equal to the record."""
    monkeypatch.delenv("GPM_MIN_CHUNK_COST", raising=False)  # the library's own chunking, as the benchmark runs it
    cfg, n, off_h, adj_h, labels = _generate(config)
    rec = FULL[str(config)]
    assert (n, int(off_h[-1])) == (rec["vertices"], rec["arcs"])
    m = binding.Matcher(1, devices=[0])
    m.upload(n, off_h, adj_h, labels=labels)
    m.set_pattern(gen.PATTERNS[cfg["pattern"]], labels=cfg.get("pattern_labels"))
    assert m.findMatches(count_only=True) == rec["embeddings"] == bench.EXPECTED_FULL[config]
    st = m.stats()
    names = [binding.hip_lib().gpm_hip_kernel_name(k).decode() for k in range(binding.KINDS)]
    assert st.kernel_launches[names.index(KIND_NAMES[config])] > 0
    if config == 1:  # a graph smaller than one launch's worth of work is one chunk
        assert st.chunks[0] == 1
    m.close()


def test_configuration_4_through_the_program(tmp_path):
    """Configuration 4 (power-law graph of 2^22 vertices, 16 vertex labels, labelled 5-vertex tree) as an input
    file with the labels trailer through gpm-par.exe, count-only: loader, upload, planner and search of the
    drop-in program at full size.  The two stdout lines are the reference's (reference: src/gpm-par.c:583, :626)."""
    cfg, n, off_h, adj_h, labels = _generate(4)
    inp, out = str(tmp_path / "in.txt"), str(tmp_path / "out.txt")
    binding.write_input_csr(inp, n, off_h, adj_h, gen.PATTERNS[cfg["pattern"]], data_labels=labels,
                            pattern_labels=cfg["pattern_labels"])
    env = dict(os.environ, GPM_COUNT_ONLY="1", GPM_STATS="1")
    env.pop("GPM_MIN_CHUNK_COST", None)
    r = subprocess.run([os.path.join(util.ROOT, "gpm-par.exe"), inp, out], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 2 and lines[0].startswith("Distribution time[s]: ") and lines[1].startswith("Computations time[s]: ")
    assert f"matches: {FULL['4']['embeddings']}" in r.stderr
    assert "load_s" in r.stderr and "upload_s" in r.stderr  # what "Distribution time" is made of
    assert os.path.getsize(out) == 0
