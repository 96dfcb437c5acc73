#!/usr/bin/env python3
"""Time a pattern under several binding orders (GPM_PLAN_ORDER) on one GPU.

    python scripts/plan_orders.py <config 1..5 | pattern> [--scale S] [--orders "1,3,5,2,4,6;default;reference"]
"""
import argparse
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mpi_amd import binding, gen  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("what")
ap.add_argument("--scale", type=int, default=16)
ap.add_argument("--ef", type=int, default=16)
ap.add_argument("--orders", default="default;reference")
ap.add_argument("--all", action="store_true", help="every connected order (small patterns)")
ap.add_argument("--budget", type=float, default=30.0, help="skip orders after this many seconds in total")
args = ap.parse_args()

if args.what.isdigit():
    cfg = bench.CONFIGS[int(args.what)]
    spec, pname = cfg["graph"], cfg["pattern"]
else:
    spec, pname = {"graph": "rmat", "scale": args.scale, "ef": args.ef, "seed": 1}, args.what
dev = torch.device("cuda", 0)
n, off, adj = bench.generate(spec, dev)
off_h, adj_h = off.cpu().numpy(), adj.cpu().numpy()
pat = gen.PATTERNS[pname]
P, arcs = pat
nbr = {v: set() for v in range(1, P + 1)}
for a, b in arcs:
    if a != b:
        nbr[a].add(b)
        nbr[b].add(a)
orders = [o for o in args.orders.split(";") if o]
if args.all:
    for perm in itertools.permutations(range(1, P + 1)):
        if all(any(w in nbr[perm[k]] for w in perm[:k]) for k in range(1, P)):
            orders.append(",".join(map(str, perm)))
m = binding.Matcher(1)
m.upload(n, off_h, adj_h)
spent, results = 0.0, []
for o in orders:
    if spent > args.budget:
        break
    if o == "default":
        os.environ.pop("GPM_PLAN_ORDER", None)
    else:
        os.environ["GPM_PLAN_ORDER"] = o
    plan = m.set_pattern(pat)
    m.findMatches(count_only=True)
    t0 = time.perf_counter()
    c = m.findMatches(count_only=True)
    dt = time.perf_counter() - t0
    spent += 2 * dt
    st = m.stats()
    rows = [int(x) for x in list(st.rows_in)[:P]]
    results.append((dt, o, c, rows))
    print(f"{o:24s} order {[x + 1 for x in list(plan.order)[:P]]} aut {plan.n_aut:4d} count {c} {dt * 1e3:9.2f} ms rows {rows}", flush=True)
results.sort()
print("best:", results[0][1], f"{results[0][0] * 1e3:.2f} ms")
