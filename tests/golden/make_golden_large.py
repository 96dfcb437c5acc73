#!/usr/bin/env python3
"""Capture golden vectors too large to keep as sorted-line md5s.

Runs the compiled reference (oracle/_ref/gpm-par-ref, built by `make -C oracle
ref` from the reference's own sources) under `mpirun -np 2` on generated
inputs whose outputs run to 10^8 .. 10^9 lines.  The output file is a named
pipe: a reader counts the lines and accumulates an order-independent
checksum of the rows (tests/util.py: rows_multiset_hash), so nothing of that
size is stored.  Records go to tests/golden/golden_large.json:

    spec, input md5, "count", "rows_hash", "source", "seconds"

"source" is "reference" for the compiled reference.  With --oracle, or when
the reference exceeds --budget seconds, the pinned CPU oracle (fast = 1:
binary-search membership, same matches) produces the record instead and
"source" says "oracle".  Labelled specs always come from the oracle with the
label test applied while a vertex is bound ("source": "oracle-labels"): the
reference has no labels, and its unlabelled output on these inputs (10^13
lines) cannot be enumerated and filtered.

    python tests/golden/make_golden_large.py [--only NAME ...] [--budget S] [--oracle]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpi_amd import gen  # noqa: E402
from tests import util  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "gpm-par-ref")
MPIRUN = os.environ.get("MPIRUN", "/opt/conda/bin/mpirun")
OUT = os.environ.get("GOLDEN_LARGE_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_large.json")

SPECS = [
    {"name": "rmat12_ef16_clique4", "graph": "rmat", "scale": 12, "ef": 16, "seed": 1, "pattern": "clique4"},
    {"name": "rmat14_ef16_clique4", "graph": "rmat", "scale": 14, "ef": 16, "seed": 1, "pattern": "clique4"},
    {"name": "rmat11_ef16_cycle4", "graph": "rmat", "scale": 11, "ef": 16, "seed": 1, "pattern": "cycle4"},
    {"name": "rmat12_ef16_cycle4", "graph": "rmat", "scale": 12, "ef": 16, "seed": 1, "pattern": "cycle4"},
    {"name": "skew10_k6_minus_pm", "graph": "rmat", "scale": 10, "ef": 8, "abcd": [0.7, 0.1, 0.1, 0.1], "seed": 1,
     "pattern": "k6_minus_pm"},
    {"name": "rmat12_ef16_triangle", "graph": "rmat", "scale": 12, "ef": 16, "seed": 1, "pattern": "triangle"},
    # directed inputs (each sampled arc kept as u -> v, no reverse arc): the in-CSR differs from the out-CSR
    {"name": "rmatdir12_ef8_tri_dir", "graph": "rmat", "scale": 12, "ef": 8, "seed": 1, "directed": True, "pattern": "tri_dir"},
    {"name": "rmatdir12_ef8_k4_dag", "graph": "rmat", "scale": 12, "ef": 8, "seed": 1, "directed": True, "pattern": "k4_dag"},
    {"name": "rmatdir12_ef8_rev4", "graph": "rmat", "scale": 12, "ef": 8, "seed": 1, "directed": True, "pattern": "rev4"},
    {"name": "rmatdir9_ef8_tree5_dir", "graph": "rmat", "scale": 9, "ef": 8, "seed": 1, "directed": True, "pattern": "tree5_dir"},
    # a directed multigraph: a quarter of its arcs are listed more than once, and a candidate listed twice is
    # tried twice (reference: src/gpm-par.c:212-213)
    {"name": "rmatmulti10_ef16_tri_dir", "graph": "rmat", "scale": 10, "ef": 16, "seed": 1, "directed": True, "multi": True,
     "pattern": "tri_dir"},
    {"name": "rmatmulti10_ef16_k4_dag", "graph": "rmat", "scale": 10, "ef": 16, "seed": 1, "directed": True, "multi": True,
     "pattern": "k4_dag"},
    {"name": "rmatmulti10_ef16_rev4", "graph": "rmat", "scale": 10, "ef": 16, "seed": 1, "directed": True, "multi": True,
     "pattern": "rev4"},
    # config 4's twin: 16 labels on the n = 2^14 power-law graph with the full configuration's average degree
    {"name": "plaw14_tree5_labelled16", "graph": "plaw", "n": 1 << 14, "avg_deg": 16, "seed": 1, "pattern": "tree5",
     "labels": 16, "pattern_labels": [0, 1, 2, 3, 1]},
    # ... and one the reference can enumerate unlabelled (sparser, smaller), filtered by label afterwards
    {"name": "plaw11_tree5_labelled16", "graph": "plaw", "n": 1 << 11, "avg_deg": 3, "seed": 1, "pattern": "tree5",
     "labels": 16, "pattern_labels": [0, 1, 2, 3, 1]},
]


def run_reference(inp, ncols, budget, keep=None):
    """(count, hash, seconds) of the reference's output, or None when it runs out of budget.
    keep(rows) -> mask filters the rows that count (label post-filter)."""
    with tempfile.TemporaryDirectory() as tmp:
        fifo = os.path.join(tmp, "out.fifo")
        os.mkfifo(fifo)
        result = {}

        def reader():
            with open(fifo, "rb") as f:
                if keep is None:
                    result["v"] = util.text_multiset_hash(f, ncols)
                    return
                count, total, carry = 0, 0, b""
                while True:
                    buf = f.read(1 << 26)
                    if not buf:
                        break
                    buf = carry + buf
                    cut = buf.rfind(b"\n") + 1
                    carry, buf = buf[cut:], buf[:cut]
                    if not buf:
                        continue
                    rows = np.fromstring(buf.decode(), dtype=np.int64, sep=" ").reshape(-1, ncols)
                    rows = rows[keep(rows)]
                    count += len(rows)
                    total = (total + util.rows_multiset_hash(rows)) & ((1 << 64) - 1)
                result["v"] = (count, total)

        th = threading.Thread(target=reader)
        th.start()
        t0 = time.time()
        # the reference opens its output with "w+" (reference: src/gpm-par.c:62): a pipe opened for
        # reading and writing does not block, and truncation is a no-op on it
        proc = subprocess.Popen([MPIRUN, "-np", "2", REF, inp, fifo], stdout=subprocess.DEVNULL)
        try:
            proc.wait(timeout=budget)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()
            # unblock the reader
            with open(fifo, "wb"):
                pass
            th.join()
            return None
        dt = time.time() - t0
        with open(fifo, "wb"):  # in case the writer never opened it
            pass
        th.join()
        if proc.returncode != 0:
            raise RuntimeError(f"reference exited with {proc.returncode}")
        return result["v"][0], result["v"][1], dt


def run_oracle(spec, n, s, d, dl, pl):
    """(count, hash, seconds) from the CPU oracle, rows taken in slices of the root range."""
    pat = gen.PATTERNS[spec["pattern"]]
    t0 = time.time()
    count, total = 0, 0
    step = max(1, n // 256)
    for lo in range(1, n + 1, step):
        hi = min(n, lo + step - 1)
        c, rows, _ = util.oracle_matches(n, s, d, pat, fast=2 if dl is not None else 1, data_labels=dl,
                                         pattern_labels=pl, root_lo=lo, root_hi=hi)
        count += c
        total = (total + util.rows_multiset_hash(rows)) & ((1 << 64) - 1)
    return count, total, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--budget", type=int, default=3600)
    ap.add_argument("--oracle", action="store_true")
    args = ap.parse_args()
    for spec in SPECS:
        if args.only and spec["name"] not in args.only:
            continue
        n, s, d = util.build_graph(spec)
        dl, pl = util.build_labels(spec, n)
        P = gen.PATTERNS[spec["pattern"]][0]
        with tempfile.TemporaryDirectory() as tmp:
            inp = os.path.join(tmp, "in.txt")
            # the reference gets the unlabelled input; labels are applied to its output
            gen.write_input(inp, n, s, d, gen.PATTERNS[spec["pattern"]])
            with open(inp, "rb") as f:
                input_md5 = hashlib.md5(f.read()).hexdigest()
            res, source = None, None
            big_labelled = dl is not None and n > 4096
            if not args.oracle and not big_labelled:
                keep = None
                if dl is not None:
                    lab = np.asarray(dl)
                    want = np.asarray(pl)
                    keep = lambda rows: (lab[rows - 1] == want[None, :]).all(axis=1)  # noqa: E731
                res = run_reference(inp, P, args.budget, keep)
                source = "reference" if dl is None else "reference, output filtered by label"
            if res is None:
                if big_labelled:
                    res = run_oracle(spec, n, s, d, dl, pl)
                    source = "oracle-labels"
                else:
                    res = run_oracle(spec, n, s, d, dl, pl)
                    source = "oracle"
        rec = dict(spec)
        rec.update({"input_md5": input_md5, "count": int(res[0]), "rows_hash": int(res[1]), "source": source,
                    "seconds": round(res[2], 1)})
        print(json.dumps(rec), flush=True)
        golden = []
        if os.path.exists(OUT):
            with open(OUT) as f:
                golden = json.load(f)
        golden = [g for g in golden if g["name"] != spec["name"]] + [rec]
        golden.sort(key=lambda g: g["name"])
        with open(OUT, "w") as f:
            json.dump(golden, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
