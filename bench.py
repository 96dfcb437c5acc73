#!/usr/bin/env python3
"""Headline benchmark: embeddings of the 4-clique found per second on an R-MAT
scale-24 graph (edge factor 16), count-only, on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  Started from a plain shell with N > 1, the script
launches its N ranks itself (`python -m torch.distributed.run`, rendezvous on
127.0.0.1) before anything touches a GPU; started by torch.distributed.run it
is one of the ranks.  Every rank builds the same synthetic graph on its own
GPU (deterministic generator), uploads it through the C ABI, and all ranks
take root chunks from one shared cursor; the only exchange is an all-reduce
of the counts (RCCL when N > 1).  A step is one complete search of the graph;
the graph is resident in HBM before the timed region.  Rank 0 prints one JSON
line and exits non-zero when the count is not the expected one.
"""
import argparse
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
VALU_PEAK_GINSTR = 256 * 2.4  # wave-level vector-ALU instructions per ns: 256 CUs x 1 per clock x 2.4 GHz


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# The five benchmark configurations: data graph, pattern, and the scaled-down twin the CPU baseline runs on.
CONFIGS = {
    1: {"graph": {"graph": "er", "n": 10000, "m": 50000, "seed": 1}, "pattern": "triangle",
        "twin": {"graph": "er", "n": 10000, "m": 50000, "seed": 1}, "baseline_ranks": 4},
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

# Embeddings of every configuration at full size.  No CPU program can enumerate them.  All five are pinned by
# independent counts on the CPU (tests/golden/full_size_counts.json): triangles, 4-cliques, 4-cycles and
# octahedra (K6 minus a perfect matching) by oracle/clique_count.c, the labelled tree by the closed form of
# tests/golden/tree_count.py -- checkers that share nothing with the HIP library and are themselves pinned to
# the reference's vectors (tests/test_clique_count.py, tests/test_tree_count.py).
EXPECTED_FULL = {1: 924, 2: 3614591580112, 3: 21123390052368, 4: 20278778911046, 5: 235455970772256}
EXPECTED_FULL_SOURCE = {
    1: "independent CPU count (oracle/clique_count.c), tests/golden/full_size_counts.json",
    2: "independent CPU count (oracle/clique_count.c), tests/golden/full_size_counts.json",
    3: "independent CPU count (oracle/clique_count.c), tests/golden/full_size_counts.json",
    4: "independent closed-form count on the CPU (tests/golden/tree_count.py), tests/golden/full_size_counts.json",
    5: "independent CPU count (oracle/clique_count.c), tests/golden/full_size_counts.json",
}


def spec_key(graph, pattern, labels=None, pattern_labels=None):
    g = {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in graph.items()}
    return json.dumps({"graph": g, "pattern": pattern, "labels": labels,
                       "pattern_labels": list(pattern_labels) if pattern_labels else None}, sort_keys=True)


def expected_count(config, cfg, overridden, labelled):
    """(count, where it comes from) for the workload about to run, or (None, None)."""
    if not overridden:
        return EXPECTED_FULL[config], EXPECTED_FULL_SOURCE.get(
            config, "count the device paths agree on at full size (bench.py EXPECTED_FULL)")
    want = spec_key(cfg["graph"], cfg["pattern"], cfg.get("labels") if labelled else None,
                    cfg.get("pattern_labels") if labelled else None)
    for name in ("golden.json", "golden_large.json"):
        path = os.path.join(ROOT, "tests", "golden", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            for g in json.load(f):
                graph = {k: g[k] for k in ("graph", "n", "m", "scale", "ef", "abcd", "avg_deg", "seed", "p", "symmetric")
                         if k in g}
                if spec_key(graph, g["pattern"], g.get("labels"), g.get("pattern_labels")) == want:
                    return g["count"], f"tests/golden/{name}: {g['name']} ({g.get('source', 'reference')})"
    return None, None


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
    from mpi_amd import gen
    if spec["graph"] == "er":
        return spec["n"], *gen.erdos_renyi(spec["n"], spec["m"], seed=spec["seed"], device=device)
    if spec["graph"] == "rmat":
        return 1 << spec["scale"], *gen.rmat(spec["scale"], spec["ef"], abcd=tuple(spec.get("abcd", (0.57, 0.19, 0.19, 0.05))),
                                             seed=spec["seed"], device=device)
    return spec["n"], *gen.power_law(spec["n"], avg_deg=spec["avg_deg"], seed=spec["seed"], device=device)


# ---- CPU baseline -----------------------------------------------------------

def host_cores():
    """(cores the machine reports, cores this job may use).  On a shared box the second is the smaller:
    thread-count variables set by the operator bound it."""
    total = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else total
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            usable = min(usable, int(v))
    return total, usable


def count_lines(path):
    with open(path, "rb") as f:
        return sum(chunk.count(b"\n") for chunk in iter(lambda: f.read(1 << 24), b""))


def run_reference(binary, ranks, inp, out, budget_s):
    """Wall seconds of `mpirun -np <ranks> <binary> inp out`, None when it ran out of budget."""
    mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
    t0 = time.time()
    try:
        subprocess.run([mpirun, "-np", str(ranks), binary, inp, out], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=budget_s)
    except subprocess.TimeoutExpired:
        return None
    return time.time() - t0


def cpu_baseline(cfg, budget_s=120, many_budget_s=60, forced_many=0):
    """The reference (built from its own sources into oracle/_ref) timed beside the GPU run on a scaled-down
    twin of the workload, full output to a file:
      * `mpirun -np 2` -- one worker, its fastest setting (it scales negatively: every candidate a worker does
        not own costs an MPI round trip, reference: src/gpm-par.c:223-229); configuration 1 uses `-np 4`, the
        setting its definition names;
      * `mpirun -np K`, K = min(cores this job may use, 64), with the build that keeps the payload of its
        fire-and-forget sends alive (oracle/ref_isend_fix.c; the untouched reference crashes at K >= 3),
        under a budget: "DNF" when it does not finish.
    Falls back to the oracle (a port) when the reference binary or mpirun is missing, and for labelled
    patterns, which the reference cannot read."""
    import numpy as np
    from mpi_amd import gen
    from tests import util
    pattern_name = cfg["pattern"]
    n, off, adj = generate(cfg["twin"], "cpu")
    s, d = gen.edges_from_csr(*gen.csr_to_numpy(off, adj))
    pat = gen.PATTERNS[pattern_name]
    dl = pl = None
    if "labels" in cfg:
        dl = gen.vertex_labels(n, cfg["labels"], seed=cfg["twin"]["seed"]).numpy().astype(np.int32)
        pl = cfg["pattern_labels"]
    total_cores, usable = host_cores()
    sample = f"{describe(cfg['twin'])} twin of the workload, {pattern_name}, full output to a file"
    ref = os.path.join(ROOT, "oracle", "_ref", "gpm-par-ref")
    ref_k = os.path.join(ROOT, "oracle", "_ref", "gpm-par-ref-k")
    mpirun = shutil.which("mpirun") or "/opt/conda/bin/mpirun"
    with tempfile.TemporaryDirectory() as tmp:
        inp, out = os.path.join(tmp, "in.txt"), os.path.join(tmp, "out.txt")
        gen.write_input(inp, n, s, d, pat, dl, pl)
        if dl is None and os.path.exists(ref) and os.path.exists(mpirun):
            ranks = int(cfg.get("baseline_ranks", 2))
            binary = ref if ranks == 2 else ref_k
            dt = run_reference(binary, ranks, inp, out, budget_s) if os.path.exists(binary) else None
            if dt is not None:
                count = count_lines(out)
                base = {"value": count / dt, "unit": "matches/s", "cores": ranks, "kind": "reference",
                        "sample": sample + f", mpirun -np {ranks} ({ranks - 1} worker(s) + coordinator)",
                        "matches": count, "seconds": round(dt, 3), "host_cores": total_cores,
                        "host_cores_usable": usable}
                legs = {}
                if ranks != 2:  # also its fastest setting
                    dt2 = run_reference(ref, 2, inp, out, budget_s)
                    legs["np2"] = ({"value": count_lines(out) / dt2, "cores": 2, "seconds": round(dt2, 3)}
                                   if dt2 is not None else {"value": None, "cores": 2, "note": f"DNF in {budget_s} s"})
                many = forced_many if forced_many >= 3 else min(usable, 64)
                if many >= 3 and many != ranks and os.path.exists(ref_k) and many_budget_s > 0:
                    dtk = run_reference(ref_k, many, inp, out, many_budget_s)
                    legs[f"np{many}"] = ({"value": count_lines(out) / dtk, "cores": many, "seconds": round(dtk, 3)}
                                         if dtk is not None else
                                         {"value": None, "cores": many, "note": f"DNF in {many_budget_s} s"})
                if legs:
                    base["other_rank_counts"] = legs
                return base
            log("reference baseline unavailable; timing the oracle port instead")
        lib = util.oracle_lib()
        t0 = time.time()
        count = lib.oracle_run_file(inp.encode(), out.encode(), 0)
        dt = time.time() - t0
        if dl is not None:
            sample += " (the reference has no labels: every unlabelled match is enumerated, then filtered)"
        return {"value": count / dt, "unit": "matches/s", "cores": 1, "kind": "port", "sample": sample,
                "matches": int(count), "seconds": round(dt, 3), "host_cores": total_cores, "host_cores_usable": usable}


# ---- launch -----------------------------------------------------------------

def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """Started from a plain shell with --gpus N > 1: run N ranks of this script under torch.distributed.run
    as a child (nothing in this process has touched a GPU), pass its output through, exit with its code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def kernels_fingerprint():
    h = hashlib.sha256()
    for name in ("gpm_kernels.hpp", "gpm_device.hpp", "gpm_hip.hip"):
        with open(os.path.join(ROOT, "mpi_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="benchmark configuration; 3 (R-MAT scale-24 4-clique) is the headline")
    ap.add_argument("--scale", type=int, default=None, help="override the R-MAT scale of the configuration")
    ap.add_argument("--ef", type=int, default=None, help="override the R-MAT edge factor of the configuration")
    ap.add_argument("--pattern", default=None, help="override the pattern of the configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-search", action="store_true",
                    help="skip the untimed searches after the timed region (byte accounting, one chunk in flight); "
                         "profiling runs: their launches would mix into the per-kernel averages of the timed steps")
    ap.add_argument("--cpu-ranks", type=int, default=0,
                    help="rank count of the many-rank leg of the CPU baseline (default min(usable cores, 64))")
    ap.add_argument("--cpu-many-budget", type=int, default=60, help="seconds the many-rank leg may take (0 = skip)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    from mpi_amd import binding, dist as gdist, gen

    cfg = dict(CONFIGS[args.config])
    cfg["graph"] = dict(cfg["graph"])
    overridden = False
    if cfg["graph"]["graph"] == "rmat":
        if args.scale is not None and args.scale != cfg["graph"]["scale"]:
            cfg["graph"]["scale"] = args.scale
            overridden = True
        if args.ef is not None and args.ef != cfg["graph"]["ef"]:
            cfg["graph"]["ef"] = args.ef
            overridden = True
    if args.pattern is not None and args.pattern != cfg["pattern"]:
        cfg["pattern"] = args.pattern
        overridden = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # GPM_BENCH_BACKEND=gloo is for rehearsing N > 1 on a box with fewer GPUs than ranks
    backend = os.environ.get("GPM_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
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
    labelled = "labels" in cfg and args.pattern is None
    if labelled:
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
    # Root chunks of equal estimated cost, shared by all ranks.  Finer chunks balance the ranks better at the
    # end of a search but cost GPU time (smaller launches: 64 -> 191.6 ms, 128 -> 192.9, 256 -> 198.5,
    # 512 -> 208.2 for the whole 4-clique search on one GPU), so the count grows more slowly than the ranks.
    matcher.set_chunks(int(os.environ.get("GPM_BENCH_CHUNKS", 64 * world if world <= 2 else 32 * world)))
    # timed steps run without the byte accounting of the clique kernels (an add per adjacent pair inside the
    # kernels); the bytes, a property of the work, come from one more search after the timed region
    matcher.set_accounting(False)
    kinds = binding.KINDS
    k_ms, k_launches = [0.0] * kinds, [0] * kinds
    timing = {"on": False}

    def step():
        # the all-reduce of the counts that ends every step is its only collective: the two cursor words
        # alternate, so no barrier is needed around a reset
        matcher.set_cursor(gdist.begin_step(cursor))
        local = matcher.findMatches(count_only=True)
        st = matcher.stats()  # HIP-event times of every launch, on the streams the kernels ran on
        if timing["on"]:
            for k in range(kinds):
                k_ms[k] += st.kernel_ms[k]
                k_launches[k] += st.kernel_launches[k]
        return gdist.all_reduce_count(local, device=coll_dev), st.ms_total, int(st.chunks[0])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timing["on"] = True

    elapsed, total, busy, taken = gdist.timed_steps(step, args.steps, args.warmup, fence, device=coll_dev)
    timing["on"] = False
    st = matcher.stats()
    sec_per_step = elapsed / max(args.steps, 1)

    # Outside the timed region, rank 0 alone (its own cursor word, every chunk): one search with the byte
    # accounting on -- the merge-model bytes of every kernel family -- and, for the clique path, one with a
    # single chunk in flight, where every launch has the GPU to itself.
    k_bytes, alone = [0] * kinds, None
    if rank == 0 and not args.no_extra_search:
        matcher.set_cursor(0)
        matcher.set_accounting(True)
        check = matcher.findMatches(count_only=True)
        sb = matcher.stats()
        k_bytes = [int(x) for x in sb.kernel_alg_bytes]
        if world == 1 and check != total:
            raise SystemExit(f"the accounting search counted {check}, the timed steps {total}")
        matcher.set_accounting(False)
        if world == 1 and (cfg["pattern"].startswith("clique") or cfg["pattern"] == "triangle"):
            os.environ["GPM_CLIQUE_INFLIGHT"] = "1"
            t1 = time.perf_counter()
            matcher.findMatches(count_only=True)
            alone_ms = (time.perf_counter() - t1) * 1e3
            sa = matcher.stats()
            alone = {"ms_per_step": alone_ms, "kernel_ms": list(sa.kernel_ms), "kernel_launches": list(sa.kernel_launches)}
            del os.environ["GPM_CLIQUE_INFLIGHT"]

    rc = 0
    if rank == 0:
        steps = max(args.steps, 1)
        names = [binding.hip_lib().gpm_hip_kernel_name(k).decode() for k in range(kinds)]
        # The roofline is quoted on the kernel family with the most device time in the timed steps, whatever
        # it is.  Bytes per step are this rank's share of the accounting search's.
        dom = max(range(kinds), key=lambda k: k_ms[k])
        share = 1.0  # of the family's launches of a whole search, the part this rank ran per step
        if world > 1 and not args.no_extra_search:
            share = (k_launches[dom] / steps) / max(int(sb.kernel_launches[dom]), 1)
        tj, counters_note = {}, "no committed counter pass"
        # the counter passes are of one GPU running a configuration at its own size: hbm_traffic.json for the
        # headline workload, config<N>_hbm_traffic.json for the others
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json" if args.config == 3 else f"config{args.config}_hbm_traffic.json")
        if overridden or world > 1:
            counters_note = "committed counter passes are of the single-GPU run at the configuration's size: not reported"
        elif os.path.exists(tpath):  # HBM bytes and vector-ALU instructions per launch from the committed PMC passes
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("kernels_fingerprint") == kernels_fingerprint():
                counters_note = (f"committed builder-side rocprofv3 counter passes of these kernel sources "
                                 f"({tj.get('source', tpath)}); times are this run's HIP events")
            else:
                tj = {}
                counters_note = "committed counter pass is of other kernel sources (stale): not reported"

        def family(k):
            """Per launch of family k: time (HIP events; without overlap where the extra search measured it),
            counter-derived use of the vector ALUs and of HBM, and the merge-model rate as a side figure."""
            if not k_launches[k]:
                return None
            avg_ms = k_ms[k] / k_launches[k]
            fam = {"ms_per_step": k_ms[k] / steps, "launches_per_step": k_launches[k] / steps, "avg_launch_ms": avg_ms}
            t_ms, t_from = avg_ms, "timed steps (launches of up to four chunks overlap)" if alone else "timed steps"
            if alone and alone["kernel_launches"][k]:
                t_ms = alone["kernel_ms"][k] / alone["kernel_launches"][k]
                t_from = "extra search with one chunk in flight (launches do not overlap)"
                fam["avg_launch_ms_alone"] = t_ms
            fam["rates_over"] = t_from
            c = tj.get(names[k], {})
            if c.get("valu_instructions_per_launch") and t_ms > 0:
                fam["valu_Ginstr_per_launch"] = c["valu_instructions_per_launch"] / 1e9
                fam["valu_Ginstr_per_s"] = c["valu_instructions_per_launch"] / 1e9 / (t_ms / 1e3)
                fam["valu_frac"] = fam["valu_Ginstr_per_s"] / VALU_PEAK_GINSTR
            if c.get("hbm_bytes_per_launch") is not None and t_ms > 0:
                fam["hbm_GB_per_launch"] = c["hbm_bytes_per_launch"] / 1e9
                fam["hbm_GBps"] = c["hbm_bytes_per_launch"] / 1e9 / (t_ms / 1e3)
                fam["hbm_frac"] = fam["hbm_GBps"] / HBM_PEAK_GBS
            if c.get("lds_conflict_share") is not None:
                fam["lds_conflict_share"] = c["lds_conflict_share"]
            if k_bytes[k] and t_ms > 0:  # what streaming every operand list once per partial match would move
                per_launch = k_bytes[k] * (share if k == dom else 1.0) * steps / k_launches[k]
                fam["merge_model_GB_per_launch"] = per_launch / 1e9
                fam["merge_model_GBps"] = per_launch / 1e9 / (t_ms / 1e3)
                fam["merge_model_frac"] = fam["merge_model_GBps"] / HBM_PEAK_GBS
            return fam

        families = {names[k]: family(k) for k in range(kinds) if k_launches[k]}
        fd = families[names[dom]]
        # which resource binds the dominant family: the larger of its counter-derived shares; without counters
        # the merge model stands in, as long as it prices the kernel below the HBM peak
        roof = {"kernel": names[dom], "counters_from": counters_note, "launches": k_launches[dom],
                "avg_launch_ms": fd.get("avg_launch_ms_alone", fd["avg_launch_ms"]), "rates_over": fd["rates_over"],
                "traffic": tj.get(names[dom], {}).get("hbm_bytes_per_launch"),
                "merge_model_frac": fd.get("merge_model_frac"), "merge_model_GBps": fd.get("merge_model_GBps"),
                "alg_bytes_per_launch": None if "merge_model_GB_per_launch" not in fd else fd["merge_model_GB_per_launch"] * 1e9}
        if fd.get("valu_frac", 0.0) >= fd.get("hbm_frac", 0.0) and "valu_frac" in fd:
            roof.update({"bound": "valu", "achieved": fd["valu_Ginstr_per_s"], "peak": VALU_PEAK_GINSTR, "unit": "Ginstr/s",
                         "frac": fd["valu_frac"], "hbm_frac": fd.get("hbm_frac"),
                         "note": "the family is bound by vector-ALU issue: wave-level VALU instructions per launch "
                                 "(SQ_INSTS_VALU) over the launch time, against 256 CUs x 1 instruction per clock x "
                                 "2.4 GHz; hbm_frac is the rate of the bytes HBM really moved (FETCH_SIZE, WRITE_SIZE) "
                                 "against 8 TB/s; merge_model_frac prices the list streaming the kernel avoids and is "
                                 "not a utilisation"})
        elif "hbm_frac" in fd:
            roof.update({"bound": "hbm", "achieved": fd["hbm_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": fd["hbm_frac"], "valu_frac": fd.get("valu_frac"),
                         "note": "bytes HBM really moved per launch (FETCH_SIZE, WRITE_SIZE, corrected) over the launch "
                                 "time against 8 TB/s; merge_model_frac is the same time against the algorithmic bytes"})
        else:
            mm = fd.get("merge_model_frac")
            roof.update({"bound": "hbm", "achieved": fd.get("merge_model_GBps"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": mm if mm is not None and mm <= 1.0 else None,
                         "note": "no counter pass of these kernel sources is committed: merge-model bytes over the launch "
                                 "time; a model rate above the HBM peak is not a utilisation and is left out"})
        want, want_src = expected_count(args.config, cfg, overridden, labelled)
        if os.environ.get("GPM_EMULATE_RANKS"):  # this process takes every N-th chunk only: a share, not the count
            want, want_src = None, "one emulated rank's share of the chunks (GPM_EMULATE_RANKS): no expected count"
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
            "expected_matches": want,
            "expected_from": want_src,
            "count_ok": None if want is None else bool(total == want),
            "roofline": roof,
            "kernel_families_rank0": families,
            "whole_search": None if not tj.get("valu_wave_instructions_per_search") or args.config != 3 or overridden else {
                "note": "all kernels of one search, from the committed counter passes of these kernel sources: "
                        "wave-level vector-ALU instructions against 256 CUs x 1 instruction per clock x 2.4 GHz, "
                        "and HBM bytes moved against the nominal peak, both over this run's time per step",
                "valu_wave_instructions": tj["valu_wave_instructions_per_search"],
                "valu_issue_frac": tj["valu_wave_instructions_per_search"] / (256 * 2.4e9 * world) / sec_per_step,
                "hbm_GB": tj.get("hbm_bytes_per_search", 0.0) / 1e9,
                "hbm_frac": tj.get("hbm_bytes_per_search", 0.0) / 1e9 / sec_per_step / HBM_PEAK_GBS / world},
            "one_chunk_in_flight": None if alone is None else {
                "note": "one extra search outside the timed region with launches not overlapping: family times add up "
                        "to the search time",
                "ms_per_step": alone["ms_per_step"],
                "kernel_ms": {names[k]: alone["kernel_ms"][k] for k in range(kinds) if alone["kernel_launches"][k]}},
            "partial_matches_per_level_rank0": [int(x) for x in list(st.rows_in)[:plan.n_levels]],
            "search_ms_per_step_by_rank": [round(b / steps, 3) for b in busy],
            "chunks_per_step_by_rank": [round(c / steps, 2) for c in taken],
            "collective": {"backend": backend if world > 1 else None,
                           "rccl": bool(world > 1 and backend == "nccl"),
                           "per_step": "1 all-reduce of a 64-bit count (two cursor words alternate: no barrier)" if world > 1 else "none"},
            "host_cores": host_cores()[0],
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, many_budget_s=args.cpu_many_budget, forced_many=args.cpu_ranks)
            # beside the reference on its sample: what an independent counting program needed on the host for
            # this configuration at FULL size (a recorded run, not timed here; it counts, the reference and the
            # GPU's output mode enumerate)
            full_path = os.path.join(ROOT, "tests", "golden", "full_size_counts.json")
            if not overridden and os.path.exists(full_path):
                with open(full_path) as f:
                    rec = json.load(f)["configs"].get(str(args.config))
                if rec and rec.get("seconds"):
                    line["cpu_baseline"]["independent_count_full_size"] = {
                        "seconds": rec["seconds"], "threads": rec.get("threads", 1),
                        "matches_per_s": rec["embeddings"] / rec["seconds"],
                        "program": rec.get("counter", "oracle/clique_count.c"),
                        "note": "recorded by tests/golden/make_full_size_counts.py on the GPU box's host cores"}
        print(json.dumps(line), flush=True)
        if want is not None and total != want:
            log(f"COUNT MISMATCH: counted {total}, expected {want} ({want_src})")
            rc = 3

    matcher.close()
    if world > 1:
        flag = torch.tensor([rc], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        rc = int(flag.item())
        dist.barrier()
    cursor.close()
    if world > 1:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
