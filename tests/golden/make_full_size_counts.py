"""Counts of the benchmark's clique, 4-cycle and labelled-tree configurations at FULL size from independent counters
(oracle/clique_count.c: degree orientation + bit rows on the CPU, no code or layout shared with the HIP
library), written to tests/golden/full_size_counts.json.  The reference cannot run these sizes (its 500 MB
per rank budget, reference: src/gpm-par.c:303-337, and days of run time); the counter is pinned to the
reference on every triangle / 4-clique vector the reference did produce (tests/test_clique_count.py).

Run where the graphs can be generated quickly (a GPU box: the generator is the one bench.py uses, and gives
the same graph on any device); the count itself runs on the host's cores:

    python tests/golden/make_full_size_counts.py [threads] [out.json]
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpi_amd import gen  # noqa: E402
import bench  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_build", "clique_count")
K = {"triangle": "3", "clique4": "4", "cycle4": "c4", "k6_minus_pm": "oct"}


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "full_size_counts.json")
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    result = {"made_by": "tests/golden/make_full_size_counts.py", "counter": "oracle/clique_count.c", "configs": {}}
    for cfg_id, cfg in sorted(bench.CONFIGS.items()):
        if cfg["pattern"] == "tree5" and cfg["graph"]["graph"] == "plaw":
            # the labelled tree: closed form over the arcs that carry the two centres' labels, numpy on the host
            from tests.golden.tree_count import count_tree5
            t0 = time.time()
            graph = cfg["graph"]
            n = graph["n"]
            off, adj = gen.power_law(n, avg_deg=graph["avg_deg"], seed=graph["seed"], device=dev)
            labels = gen.vertex_labels(n, cfg["labels"], seed=graph["seed"], device=dev).cpu().numpy()
            o, a = off.cpu().numpy().astype(np.int64), adj.cpu().numpy().astype(np.int64)
            del off, adj
            print(f"config {cfg_id}: n={n} arcs={len(a)} generated in {time.time() - t0:.1f}s", flush=True)
            t1 = time.time()
            emb = count_tree5(n, o, a, labels, cfg["pattern_labels"])
            got = {"embeddings": emb, "vertices": n, "arcs": int(len(a)), "seconds": round(time.time() - t1, 1),
                   "counter": "tests/golden/tree_count.py",
                   "workload": bench.describe(graph) + f", {cfg['labels']} labels, tree5 with labels {cfg['pattern_labels']}"}
            print(f"config {cfg_id}: {got}", flush=True)
            result["configs"][str(cfg_id)] = got
            continue
        if cfg["pattern"] not in K or cfg.get("labels"):
            continue
        t0 = time.time()
        graph = cfg["graph"]
        if graph["graph"] == "rmat":
            off, adj = gen.rmat(graph["scale"], graph["ef"], abcd=tuple(graph.get("abcd", (0.57, 0.19, 0.19, 0.05))),
                                seed=graph["seed"], device=dev)
            n = 1 << graph["scale"]
        elif graph["graph"] == "er":
            off, adj = gen.erdos_renyi(graph["n"], graph["m"], seed=graph["seed"], device=dev)
            n = graph["n"]
        else:
            continue
        o, a = off.cpu().numpy().astype(np.int64), adj.cpu().numpy().astype(np.int32)
        del off, adj
        print(f"config {cfg_id}: n={n} arcs={len(a)} generated in {time.time() - t0:.1f}s", flush=True)
        with tempfile.NamedTemporaryFile(suffix=".csr", dir=os.environ.get("TMPDIR", "/tmp")) as f:
            np.array([n, len(a)], dtype=np.int64).tofile(f)
            o.tofile(f)
            a.tofile(f)
            f.flush()
            t1 = time.time()
            p = subprocess.run([BIN, f.name, str(K[cfg["pattern"]]), str(threads)], check=True, stdout=subprocess.PIPE, text=True)
        got = json.loads(p.stdout)
        got["seconds"] = round(time.time() - t1, 1)
        got["threads"] = threads
        got["workload"] = bench.describe(graph) + ", " + cfg["pattern"]
        print(f"config {cfg_id}: {got}", flush=True)
        result["configs"][str(cfg_id)] = got
    with open(out_path, "w") as f:
        json.dump(result, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
