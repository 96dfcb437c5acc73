#!/usr/bin/env python3
"""Headline benchmark: embeddings of the 4-clique found per second on an R-MAT
scale-24 graph (edge factor 16), count-only, on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Every rank builds the same synthetic graph on its own
GPU (deterministic generator), uploads it through the C ABI, and all ranks
take root chunks from one shared cursor; the only exchange is an all-reduce
of the counts (RCCL when N > 1).  A step is one complete search of the graph;
the graph is resident in HBM before the timed region.  Rank 0 prints one JSON
line.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from mpi_amd import binding, dist as gdist, gen  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# The five benchmark configurations: data graph, pattern, and the scaled-down twin the CPU baseline runs on.
CONFIGS = {
    1: {"graph": {"graph": "er", "n": 10000, "m": 50000, "seed": 1}, "pattern": "triangle",
        "twin": {"graph": "er", "n": 10000, "m": 50000, "seed": 1}},
    2: {"graph": {"graph": "rmat", "scale": 20, "ef": 16, "seed": 1}, "pattern": "cycle4",
        "twin": {"graph": "rmat", "scale": 9, "ef": 16, "seed": 1}},
    3: {"graph": {"graph": "rmat", "scale": 24, "ef": 16, "seed": 1}, "pattern": "clique4",
        "twin": {"graph": "rmat", "scale": 10, "ef": 16, "seed": 1}},
    4: {"graph": {"graph": "plaw", "n": 1 << 22, "avg_deg": 16, "seed": 1}, "pattern": "tree5", "labels": 16,
        "pattern_labels": [0, 1, 2, 3, 1], "twin": {"graph": "plaw", "n": 1 << 10, "avg_deg": 6, "seed": 1}},
    5: {"graph": {"graph": "rmat", "scale": 18, "ef": 8, "abcd": (0.7, 0.1, 0.1, 0.1), "seed": 1},
        "pattern": "k6_minus_pm",
        "twin": {"graph": "rmat", "scale": 9, "ef": 8, "abcd": (0.7, 0.1, 0.1, 0.1), "seed": 1}},
}


def describe(spec):
    if spec["graph"] == "er":
        return f"Erdos-Renyi n={spec['n']} m={spec['m']} seed {spec['seed']}"
    if spec["graph"] == "rmat":
        a, b, c, d = spec.get("abcd", (0.57, 0.19, 0.19, 0.05))
        return (f"R-MAT scale-{spec['scale']} ef-{spec['ef']} (a,b,c,d)=({a:g},{b:g},{c:g},{d:g}) seed {spec['seed']}"
                .replace("0.", "."))
    return f"power-law (exponent 2.1) n={spec['n']} avg-deg {spec['avg_deg']} seed {spec['seed']}"


def generate(spec, device):
    """CSR (offsets, adjacency) of a benchmark graph, built on `device`."""
    if spec["graph"] == "er":
        return spec["n"], *gen.erdos_renyi(spec["n"], spec["m"], seed=spec["seed"], device=device)
    if spec["graph"] == "rmat":
        return 1 << spec["scale"], *gen.rmat(spec["scale"], spec["ef"], abcd=tuple(spec.get("abcd", (0.57, 0.19, 0.19, 0.05))),
                                             seed=spec["seed"], device=device)
    return spec["n"], *gen.power_law(spec["n"], avg_deg=spec["avg_deg"], seed=spec["seed"], device=device)


def cpu_baseline_many_ranks(cfg, ranks, budget_s=150):
    """The reference under `mpirun -np <ranks>` on the same twin (its vertex partitioning and remote
    adjacency fetches at work).  Uses the build that keeps the payload of its fire-and-forget sends
    alive (oracle/ref_isend_fix.c); the untouched reference crashes with three or more ranks."""
    from tests import util  # noqa: F401  (keeps the import pattern of cpu_baseline)
    n, off, adj = generate(cfg["twin"], "cpu")
    s, d = gen.edges_from_csr(*gen.csr_to_numpy(off, adj))
    ref = os.path.join(ROOT, "oracle", "_ref", "gpm-par-ref-k")
    mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
    if "labels" in cfg or not (os.path.exists(ref) and os.path.exists(mpirun)):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        inp, out = os.path.join(tmp, "in.txt"), os.path.join(tmp, "out.txt")
        gen.write_input(inp, n, s, d, gen.PATTERNS[cfg["pattern"]])
        t0 = time.time()
        try:
            subprocess.run([mpirun, "-np", str(ranks), ref, inp, out], check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=budget_s)
        except subprocess.TimeoutExpired:
            return {"value": None, "unit": "matches/s", "cores": ranks, "kind": "reference",
                    "sample": f"{describe(cfg['twin'])}, {cfg['pattern']}: did not finish in {budget_s} s"}
        except Exception as e:  # noqa: BLE001
            log(f"reference at {ranks} ranks failed: {e!r}")
            return None
        dt = time.time() - t0
        with open(out, "rb") as f:
            count = sum(chunk.count(b"\n") for chunk in iter(lambda: f.read(1 << 24), b""))
        return {"value": count / dt, "unit": "matches/s", "cores": ranks, "kind": "reference",
                "sample": f"{describe(cfg['twin'])} twin of the workload, {cfg['pattern']}, full output to a file, "
                          f"mpirun -np {ranks}", "matches": count, "seconds": round(dt, 3)}


def cpu_baseline(cfg, budget_s=120):
    """The reference (built from its own sources into oracle/_ref) timed under
    `mpirun -np 2` -- one worker, its fastest configuration -- on a scaled-down
    twin of the workload; falls back to the oracle (a port) when the reference
    binary or mpirun is not available, and for labelled patterns, which the
    reference cannot read."""
    from tests import util
    pattern_name = cfg["pattern"]
    n, off, adj = generate(cfg["twin"], "cpu")
    s, d = gen.edges_from_csr(*gen.csr_to_numpy(off, adj))
    pat = gen.PATTERNS[pattern_name]
    dl = pl = None
    if "labels" in cfg:
        dl = gen.vertex_labels(n, cfg["labels"], seed=cfg["twin"]["seed"]).numpy().astype(np.int32)
        pl = cfg["pattern_labels"]
    sample = f"{describe(cfg['twin'])} twin of the workload, {pattern_name}, full output to a file"
    with tempfile.TemporaryDirectory() as tmp:
        inp, out = os.path.join(tmp, "in.txt"), os.path.join(tmp, "out.txt")
        gen.write_input(inp, n, s, d, pat, dl, pl)
        ref = os.path.join(ROOT, "oracle", "_ref", "gpm-par-ref")
        mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
        if dl is None and os.path.exists(ref) and os.path.exists(mpirun):
            try:
                t0 = time.time()
                subprocess.run([mpirun, "-np", "2", ref, inp, out], check=True, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=budget_s)
                dt = time.time() - t0
                with open(out, "rb") as f:
                    count = sum(chunk.count(b"\n") for chunk in iter(lambda: f.read(1 << 24), b""))
                return {"value": count / dt, "unit": "matches/s", "cores": 2, "kind": "reference",
                        "sample": sample + ", mpirun -np 2 (1 worker + coordinator)", "matches": count,
                        "seconds": round(dt, 3)}
            except Exception as e:  # noqa: BLE001
                log(f"reference baseline unavailable ({e!r}); timing the oracle port instead")
        lib = util.oracle_lib()
        t0 = time.time()
        count = lib.oracle_run_file(inp.encode(), out.encode(), 0)
        dt = time.time() - t0
        if dl is not None:
            sample += " (the reference has no labels: every unlabelled match is enumerated, then filtered)"
        return {"value": count / dt, "unit": "matches/s", "cores": 1, "kind": "port", "sample": sample,
                "matches": int(count), "seconds": round(dt, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="benchmark configuration; 3 (R-MAT scale-24 4-clique) is the headline")
    ap.add_argument("--scale", type=int, default=None, help="override the R-MAT scale of the configuration")
    ap.add_argument("--pattern", default=None, help="override the pattern of the configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-search", action="store_true",
                    help="skip the untimed search with one chunk in flight (profiling runs: its launches would "
                         "mix into the per-kernel averages of the timed steps)")
    ap.add_argument("--cpu-ranks", type=int, default=0,
                    help="also time the reference under mpirun -np K (K >= 3) on the twin; off by default, it is slow")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    cfg["graph"] = dict(cfg["graph"])
    if args.scale is not None and cfg["graph"]["graph"] == "rmat":
        cfg["graph"]["scale"] = args.scale
    if args.pattern is not None:
        cfg["pattern"] = args.pattern

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # GPM_BENCH_BACKEND=gloo is for rehearsing N > 1 on a box with fewer GPUs than ranks
    backend = os.environ.get("GPM_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    t0 = time.time()
    n, off, adj = generate(cfg["graph"], dev)
    m_arcs = int(off[-1].item())
    labels = plabels = None
    if "labels" in cfg and args.pattern is None:
        labels = gen.vertex_labels(n, cfg["labels"], seed=cfg["graph"]["seed"], device=dev).cpu().numpy()
        plabels = cfg["pattern_labels"]
    off_h, adj_h = off.cpu().numpy(), adj.cpu().numpy()
    del off, adj
    torch.cuda.empty_cache()
    t_gen = time.time() - t0
    t0 = time.time()
    matcher = binding.Matcher(1, devices=[dev_index])
    info = matcher.upload(n, off_h, adj_h, labels=labels)
    del off_h, adj_h
    plan = matcher.set_pattern(gen.PATTERNS[cfg["pattern"]], labels=plabels)
    t_upload = time.time() - t0
    if rank == 0:
        log(f"graph: n={n} arcs={m_arcs} max_deg={info.max_out_deg} generate {t_gen:.1f}s upload+renumber "
            f"{t_upload:.1f}s; |Aut|={plan.n_aut}")

    cursor = gdist.open_cursor(f"bench_{os.environ.get('MASTER_PORT', os.getpid())}")
    matcher.set_cursor(cursor.address)
    matcher.set_chunks(int(os.environ.get("GPM_BENCH_CHUNKS", 64 * world)))

    def step():
        gdist.reset_cursor(cursor)
        local = matcher.findMatches(count_only=True)
        return gdist.all_reduce_count(local, device=coll_dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    total = 0
    for _ in range(args.warmup):
        total = step()
    fence()
    t0 = time.perf_counter()
    kinds = binding.KINDS
    k_ms, k_bytes, k_launches = [0.0] * kinds, [0] * kinds, [0] * kinds
    search_ms = 0.0
    for _ in range(args.steps):
        total = step()
        st = matcher.stats()  # HIP-event times of every launch, on the streams the kernels ran on
        search_ms += st.ms_total
        for k in range(kinds):
            k_ms[k] += st.kernel_ms[k]
            k_bytes[k] += st.kernel_alg_bytes[k]
            k_launches[k] += st.kernel_launches[k]
    fence()
    elapsed = time.perf_counter() - t0
    busy = [search_ms]  # per-rank time inside the search: the spread shows how well the chunks balance
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([search_ms], dtype=torch.float64, device=coll_dev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        busy = [float(p.item()) for p in parts]
    st = matcher.stats()
    sec_per_step = elapsed / max(args.steps, 1)

    # Outside the timed region: one more search with a single chunk in flight, so that every launch has
    # the GPU to itself.  In the timed steps up to four chunks overlap and a launch's duration includes
    # the time its workgroups share the compute units with other kernels.
    alone = None
    if (rank == 0 and world == 1 and CONFIGS[args.config]["pattern"].startswith("clique") and args.pattern is None
            and not args.no_extra_search):
        os.environ["GPM_CLIQUE_INFLIGHT"] = "1"
        gdist.reset_cursor(cursor)
        t1 = time.perf_counter()
        matcher.findMatches(count_only=True)
        alone_ms = (time.perf_counter() - t1) * 1e3
        sa = matcher.stats()
        alone = {"ms_per_step": alone_ms, "kernel_ms": list(sa.kernel_ms), "kernel_alg_bytes": list(sa.kernel_alg_bytes),
                 "kernel_launches": list(sa.kernel_launches)}
        del os.environ["GPM_CLIQUE_INFLIGHT"]

    if rank == 0:
        # the kernel family with the most device time is the one the roofline is quoted on
        # (among the families that take their operands from HBM: k_clq_count4 works on LDS tiles of rows that
        # k_clq_build made, its merge-model rate says nothing about HBM)
        names = [binding.hip_lib().gpm_hip_kernel_name(k).decode() for k in range(kinds)]
        dom = max((k for k in range(kinds) if names[k] != "k_clq_count4"), key=lambda k: k_ms[k])
        dom_name = binding.hip_lib().gpm_hip_kernel_name(dom).decode()
        achieved = (k_bytes[dom] / 1e9) / (k_ms[dom] / 1e3) if k_ms[dom] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from the committed PMC passes, per kernel family
            with open(tpath) as f:
                traffic = json.load(f).get(dom_name, {}).get("hbm_bytes_per_launch")
        families = {binding.hip_lib().gpm_hip_kernel_name(k).decode():
                    {"ms_per_step": k_ms[k] / max(args.steps, 1), "launches_per_step": k_launches[k] / max(args.steps, 1),
                     "alg_GB_per_step": k_bytes[k] / 1e9 / max(args.steps, 1)}
                    for k in range(kinds) if k_launches[k]}
        line = {
            "metric": "pattern matches found/sec and wall time, R-MAT scale-24 4-clique at 1/2/4/8 GPUs",
            "value": total / sec_per_step,
            "unit": "matches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": sec_per_step * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"{describe(cfg['graph'])}, symmetrised, {cfg['pattern']}"
                                   f"{' with ' + str(cfg['labels']) + ' vertex labels' if labels is not None else ''}"
                                   f", count-only", "benchmark_config": args.config,
                       "vertices": n, "arcs": m_arcs, "parallelism": f"roots sharded over {world} GPU(s), "
                       "graph replicated, shared chunk cursor"},
            "matches": total,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_per_launch": k_bytes[dom] / max(k_launches[dom], 1),
                         "avg_launch_ms": k_ms[dom] / max(k_launches[dom], 1), "launches": k_launches[dom]},
            "kernel_families_rank0": families,
            "one_chunk_in_flight": None if alone is None else {
                "note": "one extra search outside the timed region with launches not overlapping",
                "ms_per_step": alone["ms_per_step"], "kernel": dom_name,
                "achieved": (alone["kernel_alg_bytes"][dom] / 1e9) / (alone["kernel_ms"][dom] / 1e3) if alone["kernel_ms"][dom] > 0 else 0.0,
                "frac": ((alone["kernel_alg_bytes"][dom] / 1e9) / (alone["kernel_ms"][dom] / 1e3) / HBM_PEAK_GBS) if alone["kernel_ms"][dom] > 0 else 0.0,
                "avg_launch_ms": alone["kernel_ms"][dom] / max(alone["kernel_launches"][dom], 1)},
            "partial_matches_per_level_rank0": [int(x) for x in list(st.rows_in)[:plan.n_levels]],
            "search_ms_per_step_by_rank": [round(b / max(args.steps, 1), 3) for b in busy],
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
            if args.cpu_ranks >= 3:
                many = cpu_baseline_many_ranks(cfg, args.cpu_ranks)
                if many:
                    line["cpu_baseline_many_ranks"] = many
        print(json.dumps(line), flush=True)

    matcher.close()
    if world > 1:
        dist.barrier()
    cursor.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
