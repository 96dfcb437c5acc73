#!/usr/bin/env python3
"""Capture golden vectors from the compiled reference.

Runs oracle/_ref/gpm-par-ref (built by `make -C oracle ref` from the
reference's own sources) under `mpirun -np 2` -- one worker, hence a
deterministic output order -- on every fixture input, and records:

  tests/golden/small/<name>.in, <name>.out   hand-written and tiny cases
  tests/golden/golden.json                   for generated cases: generator
        spec, md5 of the input text, match count, md5 of the reference's
        output file as written, md5 of its sorted lines

Generated inputs are not stored; tests rebuild them with mpi_amd.gen (a
deterministic function of the spec) and check the input md5 first.

Labelled cases: the reference has no labels, so the golden output is its
unlabelled output filtered by label_G(m(i)) == label_P(i) in this script.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpi_amd import gen  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "gpm-par-ref")
MPIRUN = os.environ.get("MPIRUN", "/opt/conda/bin/mpirun")
HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = os.path.join(HERE, "small")

# hand-written cases: name -> input text
HAND = {
    # worked example: data 1->2,1->3,2->3,2->4,3->1,4->2,4->5 ; directed triangle
    "t1": "5\n1 2\n2\n3\n2 2\n3\n4\n3 1\n1\n4 2\n2\n5\n\n3\n1 1\n2\n2 1\n3\n3 1\n1\n",
    # same, last newline missing
    "t1_no_newline": "5\n1 2\n2\n3\n2 2\n3\n4\n3 1\n1\n4 2\n2\n5\n\n3\n1 1\n2\n2 1\n3\n3 1\n1",
    # same, vertex blocks listed out of id order
    "t1_out_of_order": "5\n4 2\n2\n5\n2 2\n3\n4\n1 2\n2\n3\n3 1\n1\n\n3\n3 1\n1\n1 1\n2\n2 1\n3\n",
    # arc 1->2 listed twice: duplicate data edges multiply matches
    "dup_edge": "3\n1 3\n2\n2\n3\n2 1\n3\n\n2\n1 1\n2\n",
    # pattern has a self loop 1->1, which the matcher ignores
    "pattern_self_loop": "3\n1 2\n2\n3\n2 1\n3\n\n2\n1 2\n1\n2\n",
    # data graph self loops are harmless
    "data_self_loop": "3\n1 2\n1\n2\n2 2\n2\n3\n3 1\n1\n\n3\n1 1\n2\n2 1\n3\n3 1\n1\n",
    # single-vertex pattern: every data vertex is a match
    "single_vertex": "4\n1 1\n2\n3 1\n4\n\n1\n",
    # every step of the plan walks an in-edge: arcs 2->1, 3->1, 3->2, 4->3
    "reverse_plan": "5\n1 2\n2\n3\n2 1\n3\n4 2\n1\n2\n5 1\n4\n\n4\n2 1\n1\n3 2\n1\n2\n4 1\n3\n",
    # set-symmetric multigraph (1->2 listed twice, 2->1 once), pattern arc 2->1: the level is reached over
    # an in-edge and walks the in-list's multiplicities, not the out-list's
    "sym_multigraph_in_edge": "2\n1 2\n2\n2\n2 1\n1\n\n2\n2 1\n1\n",
    # twin pattern (2 and 3 have the same arcs) on a graph that lists 2->4 twice: how often a match is
    # printed depends on which pattern arcs the plan walks
    "multigraph_twins": "4\n1 3\n2\n3\n4\n2 2\n4\n4\n3 1\n4\n\n4\n1 3\n2\n3\n4\n2 1\n4\n3 1\n4\n",
    # path whose vertex 1 is a leaf, data edge 1-2 listed twice one way and once the other
    "multigraph_leaf_root": "3\n1 2\n2\n2\n2 2\n1\n3\n3 1\n2\n\n3\n1 1\n2\n2 2\n1\n3\n3 1\n2\n",
    # several blank lines between the sections, trailing text after the pattern
    "blank_runs": "3\n1 2\n2\n3\n2 1\n3\n\n\n\n2\n1 1\n2\n\nthis trailer is ignored\n7 7\n",
}


def md5(b: bytes) -> str:
    return hashlib.md5(b).hexdigest()


def run_ref(text: str) -> bytes:
    with tempfile.TemporaryDirectory() as d:
        i, o = os.path.join(d, "in.txt"), os.path.join(d, "out.txt")
        with open(i, "w") as f:
            f.write(text)
        subprocess.run([MPIRUN, "-np", "2", REF, i, o], check=True, stdout=subprocess.DEVNULL, timeout=900)
        with open(o, "rb") as f:
            return f.read()


def build_generated(spec):
    """spec -> input text (must match tests/util.py:build_input)."""
    from tests.util import build_input
    return build_input(spec)


GENERATED = []
for pname in ["edge", "tri_dir", "triangle", "path3", "cycle4", "clique4", "diamond", "k4_dag", "rev4",
              "tree5", "tree5_dir", "house"]:
    GENERATED.append({"name": f"rand9_dir_{pname}", "graph": "rand", "n": 9, "p": [35, 100], "seed": 7,
                      "symmetric": False, "pattern": pname})
    GENERATED.append({"name": f"rand9_sym_{pname}", "graph": "rand", "n": 9, "p": [35, 100], "seed": 7,
                      "symmetric": True, "pattern": pname})
for pname in ["tri_dir", "triangle", "cycle4", "clique4", "k4_dag", "rev4", "tree5_dir", "diamond", "clique5",
              "k6_minus_pm", "k6_minus_e"]:
    GENERATED.append({"name": f"rand40_dir_{pname}", "graph": "rand", "n": 40, "p": [20, 100], "seed": 3,
                      "symmetric": False, "pattern": pname})
GENERATED += [
    {"name": "er10k_triangle", "graph": "er", "n": 10000, "m": 50000, "seed": 1, "pattern": "triangle"},
    {"name": "rmat8_triangle", "graph": "rmat", "scale": 8, "ef": 8, "seed": 1, "pattern": "triangle"},
    {"name": "rmat8_clique4", "graph": "rmat", "scale": 8, "ef": 8, "seed": 1, "pattern": "clique4"},
    {"name": "rmat8_cycle4", "graph": "rmat", "scale": 8, "ef": 8, "seed": 1, "pattern": "cycle4"},
    {"name": "rmat8_house", "graph": "rmat", "scale": 8, "ef": 4, "seed": 2, "pattern": "house"},
    {"name": "rmat8_clique5", "graph": "rmat", "scale": 8, "ef": 8, "seed": 1, "pattern": "clique5"},
    {"name": "rmat10_triangle", "graph": "rmat", "scale": 10, "ef": 8, "seed": 1, "pattern": "triangle"},
    {"name": "rmat10_clique4", "graph": "rmat", "scale": 10, "ef": 8, "seed": 1, "pattern": "clique4"},
    {"name": "rmat10_cycle4", "graph": "rmat", "scale": 10, "ef": 8, "seed": 1, "pattern": "cycle4"},
    {"name": "rmat12_triangle", "graph": "rmat", "scale": 12, "ef": 8, "seed": 1, "pattern": "triangle"},
    {"name": "skew9_k6_minus_pm", "graph": "rmat", "scale": 9, "ef": 8, "abcd": [0.7, 0.1, 0.1, 0.1], "seed": 1,
     "pattern": "k6_minus_pm"},
    {"name": "skew8_k6_minus_e", "graph": "rmat", "scale": 8, "ef": 8, "abcd": [0.7, 0.1, 0.1, 0.1], "seed": 1,
     "pattern": "k6_minus_e"},
    {"name": "plaw8_tree5_labelled3", "graph": "plaw", "n": 256, "avg_deg": 6, "seed": 2, "pattern": "tree5",
     "labels": 3, "pattern_labels": [0, 1, 2, 0, 1]},
    {"name": "plaw9_tree5_labelled4", "graph": "plaw", "n": 512, "avg_deg": 4, "seed": 1, "pattern": "tree5",
     "labels": 4, "pattern_labels": [0, 1, 2, 3, 1]},
]


def label_filter(out: bytes, dl, pl) -> bytes:
    keep = []
    for line in out.splitlines():
        m = [int(x) for x in line.split()]
        if all(dl[m[i] - 1] == pl[i] for i in range(len(pl))):
            keep.append(line)
    return b"".join(k + b"\n" for k in keep)


def main():
    os.makedirs(SMALL, exist_ok=True)
    for name, text in HAND.items():
        out = run_ref(text)
        with open(os.path.join(SMALL, name + ".in"), "w") as f:
            f.write(text)
        with open(os.path.join(SMALL, name + ".out"), "wb") as f:
            f.write(out)
        print(name, len(out.splitlines()), "matches", flush=True)
    gpath = os.path.join(HERE, "golden.json")
    golden = []
    if os.path.exists(gpath):  # incremental: keep records whose spec is unchanged
        with open(gpath) as f:
            golden = json.load(f)
    keys = ("input_md5", "count", "output_md5", "sorted_md5")
    done = {json.dumps({k: v for k, v in g.items() if k not in keys}, sort_keys=True) for g in golden}
    golden = [g for g in golden if any(g["name"] == s["name"] for s in GENERATED)]
    for spec in GENERATED:
        if json.dumps(spec, sort_keys=True) in done:
            continue
        golden = [g for g in golden if g["name"] != spec["name"]]
        text, dl, pl = build_generated(spec)
        out = run_ref(text)
        if dl is not None:
            out = label_filter(out, dl, pl)
        lines = out.splitlines(keepends=True)
        rec = dict(spec)
        rec.update({
            "input_md5": md5(text.encode()),
            "count": len(lines),
            "output_md5": md5(out),
            "sorted_md5": md5(b"".join(sorted(lines))),
        })
        golden.append(rec)
        print(f"{spec['name']}: {len(lines)} matches", flush=True)
        with open(gpath, "w") as f:
            json.dump(golden, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
