"""Host C code (parser, CSR builder, planner) and the drop-in boundary,
checked without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from mpi_amd import binding, gen
from tests import planref, util

PATTERN_NAMES = sorted(gen.PATTERNS)


def test_c_abi_exports_every_declared_symbol():
    lib = binding.hip_lib()
    header = open(os.path.join(util.ROOT, "include", "gpm_hip.h")).read()
    import re
    declared = sorted(set(re.findall(r"\b(gpm_hip_[a-z_]+)\s*\(", header)))
    assert declared == sorted(binding.HIP_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_missing_gpu_is_reported_not_hidden():
    lib = binding.hip_lib()
    if lib.gpm_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(binding.GpmError):
        binding.Matcher(1)


@pytest.mark.parametrize("name", util.small_cases())
def test_parser_matches_oracle_parser(name, tmp_path):
    inp, _ = util.small_paths(name)
    data, pat = binding.readGraph(inp)
    lib = util.oracle_lib()
    lib.oracle_parse_file.argtypes = [ctypes.c_char_p] + [ctypes.POINTER(ctypes.c_void_p)] * 4

    class OG(ctypes.Structure):
        _fields_ = [("n", ctypes.c_int), ("out", ctypes.POINTER(ctypes.POINTER(ctypes.c_int))),
                    ("outdeg", ctypes.POINTER(ctypes.c_int)), ("inn", ctypes.POINTER(ctypes.POINTER(ctypes.c_int))),
                    ("indeg", ctypes.POINTER(ctypes.c_int)), ("ordering", ctypes.POINTER(ctypes.c_int)),
                    ("parents", ctypes.POINTER(ctypes.c_int))]

    g, p, dl, pl = (ctypes.c_void_p() for _ in range(4))
    assert lib.oracle_parse_file(inp.encode(), ctypes.byref(g), ctypes.byref(p), ctypes.byref(dl),
                                 ctypes.byref(pl)) == 0
    og = ctypes.cast(g, ctypes.POINTER(OG)).contents
    assert og.n == data["n"]
    for v in range(og.n):
        want_out = sorted(og.out[v + 1][j] - 1 for j in range(og.outdeg[v + 1]))
        want_in = sorted(og.inn[v + 1][j] - 1 for j in range(og.indeg[v + 1]))
        assert data["out_adj"][data["out_off"][v]:data["out_off"][v + 1]].tolist() == want_out
        assert data["in_adj"][data["in_off"][v]:data["in_off"][v + 1]].tolist() == want_in
    # the plan arrays equal the oracle's restatement of the reference's
    op = ctypes.cast(p, ctypes.POINTER(OG)).contents
    o, par, ok = binding.makeGraphOrdering(pat)
    assert ok
    assert o[1:op.n + 1].tolist() == [op.ordering[k] for k in range(1, op.n + 1)]
    assert par[1:op.n + 1].tolist() == [op.parents[v] for v in range(1, op.n + 1)]
    binding.host_lib().freePattern(pat)


def test_parser_rejects_bad_input(tmp_path):
    bad = {
        "id_out_of_range": "3\n1 1\n4\n\n2\n1 1\n2\n",
        "truncated_list": "3\n1 2\n2\n",
        "listed_twice": "3\n1 1\n2\n1 1\n3\n\n2\n1 1\n2\n",
        "pattern_too_large": "3\n1 1\n2\n\n11\n1 1\n2\n",
        "garbage_header": "x\n",
    }
    for name, text in bad.items():
        p = tmp_path / (name + ".in")
        p.write_text(text)
        with pytest.raises(binding.GpmError):
            binding.readGraph(str(p))


def _read_both_ways(path, monkeypatch):
    """Parse with the single-threaded parser and with the multi-threaded fast path."""
    out = []
    for threads in ("1", "5"):
        monkeypatch.setenv("GPM_LOAD_THREADS", threads)
        monkeypatch.setenv("GPM_LOAD_MIN_BYTES", "0")
        try:
            data, pat = binding.readGraph(path)
            o, par, ok = binding.makeGraphOrdering(pat)
            binding.host_lib().freePattern(pat)
            out.append(("ok", data, o.tolist(), par.tolist(), ok))
        except binding.GpmError as e:
            out.append(("error", str(e)))
    return out


def _same(a, b):
    if a[0] != b[0]:
        return False
    if a[0] == "error":
        return a[1] == b[1]
    da, db = a[1], b[1]
    for k in ("n", "symmetric", "simple"):
        if da[k] != db[k]:
            return False
    for k in ("out_off", "out_adj", "in_off", "in_adj"):
        if not np.array_equal(da[k], db[k]):
            return False
    if (da["labels"] is None) != (db["labels"] is None):
        return False
    if da["labels"] is not None and not np.array_equal(da["labels"], db["labels"]):
        return False
    return a[2:] == b[2:]


@pytest.mark.parametrize("name", util.small_cases())
def test_threaded_loader_equals_the_parser_on_small_cases(name, monkeypatch):
    inp, _ = util.small_paths(name)
    one, many = _read_both_ways(inp, monkeypatch)
    assert _same(one, many)


@pytest.mark.parametrize("spec", [g for g in util.golden_specs() if g["name"] in (
    "rand40_dir_tree5_dir", "rand40_dir_rev4", "rmat8_house", "plaw9_tree5_labelled4", "er10k_triangle",
    "rmat10_clique4", "skew9_k6_minus_pm")], ids=lambda g: g["name"])
def test_threaded_loader_equals_the_parser_on_generated_inputs(spec, tmp_path, monkeypatch):
    text, _, _ = util.build_input(spec)
    p = tmp_path / "in.txt"
    p.write_text(text)
    one, many = _read_both_ways(str(p), monkeypatch)
    assert one[0] == "ok" and _same(one, many)


def test_threaded_loader_leaves_odd_input_to_the_parser(tmp_path, monkeypatch):
    """Inputs the fast path is not sure about parse the same way (or fail with the same message)."""
    body = "".join(f"{v} 2\n{v % 50 + 1}\n{(v + 7) % 50 + 1}\n" for v in range(1, 51))
    cases = {
        "plain": "50\n" + body + "\n2\n1 1\n2\n",
        "signed_ids": "50\n" + body.replace("\n7\n", "\n+7\n", 1) + "\n2\n1 1\n2\n",
        "trailing_blanks": "50\n" + body.replace("\n9\n", "\n9  \n", 1) + "\n2\n1 1\n2\n",
        "crlf": ("50\n" + body + "\n2\n1 1\n2\n").replace("\n", "\r\n"),
        "degree_too_small": "50\n" + body.replace("3 2\n", "3 1\n", 1) + "\n2\n1 1\n2\n",
        "degree_too_large": "50\n" + body.replace("3 2\n", "3 3\n", 1) + "\n2\n1 1\n2\n",
        "id_out_of_range": "50\n" + body.replace("\n9\n", "\n99\n", 1) + "\n2\n1 1\n2\n",
        "listed_twice": "50\n" + body + "7 1\n3\n\n2\n1 1\n2\n",
        "no_blank_line": "50\n" + body,
        "long_line": "50\n" + body.replace("\n9\n", "\n" + "0" * 70 + "9\n", 1) + "\n2\n1 1\n2\n",
        "zero_degree_block": "50\n" + body.replace("50 2\n1\n8\n", "50 0\n") + "\n2\n1 1\n2\n",
        "blank_runs_before_blocks": "50\n\n\n" + body + "\n2\n1 1\n2\n",
        "with_labels": "50\n" + body + "\n2\n1 1\n2\n\nlabels\n" + "\n".join(str(v % 3) for v in range(52)) + "\n",
    }
    for name, text in cases.items():
        p = tmp_path / (name + ".in")
        p.write_bytes(text.encode())
        one, many = _read_both_ways(str(p), monkeypatch)
        assert _same(one, many), name
    assert _read_both_ways(str(tmp_path / "plain.in"), monkeypatch)[0][0] == "ok"


def test_c_writer_writes_what_the_python_writer_writes(tmp_path):
    spec = next(g for g in util.golden_specs() if g["name"] == "plaw9_tree5_labelled4")
    text, dl, pl = util.build_input(spec)
    n, s, d = util.build_graph(spec)
    order = np.lexsort((d, s))
    off = np.zeros(n + 1, dtype=np.int64)
    np.add.at(off, np.asarray(s, dtype=np.int64) + 1, 1)
    p = tmp_path / "c.txt"
    binding.write_input_csr(str(p), n, np.cumsum(off), np.asarray(d)[order], gen.PATTERNS[spec["pattern"]], dl, pl)
    assert p.read_text() == text


@pytest.mark.parametrize("name", ["plaw9_tree5_labelled4", "rand40_dir_rev4", "rmat10_clique4"])
def test_csr_cache_gives_the_same_graph(name, tmp_path, monkeypatch):
    """GPM_CSR_CACHE=1: the second read takes the data graph from the side-car file (pattern and labels
    still come from the text); a changed input invalidates it."""
    spec = next(g for g in util.golden_specs() if g["name"] == name)
    text, _, _ = util.build_input(spec)
    p = tmp_path / "in.txt"
    p.write_text(text)
    cache_dir = tmp_path / "cache"
    cache_dir.mkdir()
    first = _read_both_ways(str(p), monkeypatch)[0]
    monkeypatch.setenv("GPM_CSR_CACHE", "1")
    monkeypatch.setenv("GPM_CSR_CACHE_DIR", str(cache_dir))
    stored = _read_both_ways(str(p), monkeypatch)[0]           # parses and stores
    cars = list(cache_dir.glob("in.txt.*.gpmcsr"))  # file name + hash of the input's path
    assert len(cars) == 1
    side_car = cars[0]
    stamp = side_car.stat().st_mtime_ns
    loaded = _read_both_ways(str(p), monkeypatch)[0]           # loads
    assert side_car.stat().st_mtime_ns == stamp                # not rewritten: it was used
    assert first[0] == "ok" and _same(first, stored) and _same(first, loaded)
    # another graph under the same name: the stale side-car must not be used
    other = next(g for g in util.golden_specs() if g["name"] == "rmat8_house")
    p.write_text(util.build_input(other)[0])
    monkeypatch.delenv("GPM_CSR_CACHE")
    want = _read_both_ways(str(p), monkeypatch)[0]
    monkeypatch.setenv("GPM_CSR_CACHE", "1")
    got = _read_both_ways(str(p), monkeypatch)[0]
    assert _same(want, got) and not _same(first, got)
    # a side-car whose contents are damaged (ids out of range, same size and header) is rejected, not used
    blob = bytearray(side_car.read_bytes())
    blob[-4:] = (2 ** 31 - 1).to_bytes(4, "little")
    side_car.write_bytes(bytes(blob))
    again = _read_both_ways(str(p), monkeypatch)[0]
    assert _same(want, again)
    # the same file name in another directory gets a side-car of its own
    other_dir = tmp_path / "elsewhere"
    other_dir.mkdir()
    q = other_dir / "in.txt"
    q.write_text(text)
    there = _read_both_ways(str(q), monkeypatch)[0]
    assert _same(first, there) and len(list(cache_dir.glob("in.txt.*.gpmcsr"))) == 2


def test_symmetric_and_simple_flags(tmp_path):
    p = tmp_path / "a.in"
    p.write_text("3\n1 2\n2\n3\n2 1\n1\n3 1\n1\n\n2\n1 1\n2\n")
    data, pat = binding.readGraph(str(p))
    assert data["symmetric"] and data["simple"]
    binding.host_lib().freePattern(pat)
    inp, _ = util.small_paths("dup_edge")
    data, pat = binding.readGraph(inp)
    assert not data["simple"] and not data["symmetric"]
    binding.host_lib().freePattern(pat)


@pytest.mark.parametrize("pname", PATTERN_NAMES)
def test_plan_order_equals_reference_order(pname):
    """ordering/parents from the C planner == the oracle's restatement of graph.c:7-36."""
    pat = gen.PATTERNS[pname]
    p = binding.make_pattern(pat)
    o, par, ok = binding.makeGraphOrdering(p)
    binding.host_lib().freePattern(p)
    lib = util.oracle_lib()
    P, arcs = pat
    ps = np.array([a for a, _ in arcs], dtype=np.int32)
    pd = np.array([b for _, b in arcs], dtype=np.int32)
    og = lib.oracle_graph_from_edges(P, ps.ctypes.data, pd.ctypes.data, len(ps))
    assert (lib.oracle_make_plan(og) == 0) == ok

    class OG(ctypes.Structure):
        _fields_ = [("n", ctypes.c_int), ("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("c", ctypes.c_void_p),
                    ("d", ctypes.c_void_p), ("ordering", ctypes.POINTER(ctypes.c_int)),
                    ("parents", ctypes.POINTER(ctypes.c_int))]

    g = ctypes.cast(og, ctypes.POINTER(OG)).contents
    assert o[1:P + 1].tolist() == [g.ordering[k] for k in range(1, P + 1)]
    assert par[1:P + 1].tolist() == [g.parents[v] for v in range(1, P + 1)]
    lib.oracle_graph_free(og)


def test_disconnected_pattern_is_rejected():
    p = binding.make_pattern((3, [(1, 2)]))
    with pytest.raises(binding.GpmError):
        binding.make_plan(p, 0)
    binding.host_lib().freePattern(p)


N_AUT = {"edge": 1, "edge_und": 2, "tri_dir": 3, "triangle": 6, "path3": 2, "star4": 6, "path4": 2, "cycle4": 8, "clique4": 24,
         "diamond": 4, "k4_dag": 1, "rev4": 1, "tree5": 2, "tree5_dir": 1, "clique5": 120, "clique6": 720, "house": 2,
         "k6_minus_pm": 48, "k6_minus_e": 48}


@pytest.mark.parametrize("pname", PATTERN_NAMES)
def test_automorphism_counts(pname):
    p = binding.make_pattern(gen.PATTERNS[pname])
    plan = binding.make_plan(p, binding.PLAN_BREAK_SYMMETRY)
    binding.host_lib().freePattern(p)
    assert plan.n_aut == N_AUT[pname]


@pytest.mark.parametrize("pname", PATTERN_NAMES)
@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("symbreak", [False, True])
def test_plan_semantics_equal_oracle(pname, symmetric, symbreak):
    """Interpreting the C planner's plan in Python gives the oracle's match multiset,
    with and without symmetry breaking."""
    n = 10 if gen.PATTERNS[pname][0] <= 5 else 9
    s, d = util.rand_digraph(n, 55 if gen.PATTERNS[pname][0] >= 5 else 40, 100, 5, symmetric)
    pat = gen.PATTERNS[pname]
    cnt, rows, _ = util.oracle_matches(n, s, d, pat, fast=1)
    flags = (binding.PLAN_SYMMETRIC_DATA if symmetric else 0) | binding.PLAN_DEGREE_FILTER
    if symbreak:
        flags |= binding.PLAN_BREAK_SYMMETRY
    p = binding.make_pattern(pat)
    plan = binding.make_plan(p, flags)
    binding.host_lib().freePattern(p)
    out, inn = planref.csr_lists(n, s, d)
    canon = planref.run_plan(plan, n, out, inn, symmetric=symmetric)
    assert len(canon) * plan.n_aut == cnt
    got = sorted(tuple(x + 1 for x in r) for r in planref.expand_automorphisms(plan, canon))
    assert got == sorted(map(tuple, rows.tolist()))


TWIN_PATTERNS = {
    "diamond": gen.PATTERNS["diamond"],
    "k23": (5, sorted({(a, b) for a in (1, 2) for b in (3, 4, 5)} | {(b, a) for a in (1, 2) for b in (3, 4, 5)})),
    "tailed_k23": (5, sorted({(1, 2), (2, 1), (1, 3), (3, 1), (2, 4), (4, 2), (3, 4), (4, 3), (2, 5), (5, 2), (3, 5), (5, 3)})),
    "dir_twins": (4, [(1, 2), (2, 3), (4, 2), (3, 4), (1, 3), (4, 1)]),
    "dir_twins_in": (4, [(2, 1), (3, 1), (2, 4), (3, 4), (1, 4)]),
}


@pytest.mark.parametrize("pname", sorted(TWIN_PATTERNS))
@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("order", ["twins_last", "reference"])
def test_twins_are_bound_last(pname, symmetric, order, monkeypatch):
    """A pair of non-adjacent vertices with identical arcs goes to the end of the plan (unless the
    reference's order is asked for); either way the plan, interpreted in Python, gives the oracle's
    match multiset."""
    if order == "reference":
        monkeypatch.setenv("GPM_PLAN_ORDER", "reference")
    pat = TWIN_PATTERNS[pname]
    P = pat[0]
    n = 10
    s, d = util.rand_digraph(n, 50, 100, 7, symmetric)
    cnt, rows, _ = util.oracle_matches(n, s, d, pat, fast=1)
    flags = (binding.PLAN_SYMMETRIC_DATA if symmetric else 0) | binding.PLAN_DEGREE_FILTER | binding.PLAN_BREAK_SYMMETRY
    p = binding.make_pattern(pat)
    plan = binding.make_plan(p, flags)
    binding.host_lib().freePattern(p)
    if order == "twins_last" and pname in ("diamond", "k23"):
        arcs = set(pat[1])
        a, b = plan.order[P - 2] + 1, plan.order[P - 1] + 1
        assert (a, b) not in arcs and (b, a) not in arcs
        assert {t for t in range(1, P + 1) if (a, t) in arcs} == {t for t in range(1, P + 1) if (b, t) in arcs}
    out, inn = planref.csr_lists(n, s, d)
    canon = planref.run_plan(plan, n, out, inn, symmetric=symmetric)
    assert len(canon) * plan.n_aut == cnt
    got = sorted(tuple(x + 1 for x in r) for r in planref.expand_automorphisms(plan, canon))
    assert got == sorted(map(tuple, rows.tolist()))


def test_labelled_plan_semantics():
    n = 12
    s, d = util.rand_digraph(n, 45, 100, 9, True)
    dl = gen.vertex_labels(n, 3, seed=4).numpy().astype(np.int32)
    pl = [0, 1, 2, 1, 1]
    pat = gen.PATTERNS["tree5"]
    cnt, rows, _ = util.oracle_matches(n, s, d, pat, data_labels=dl, pattern_labels=pl)
    p = binding.make_pattern(pat, labels=pl)
    plan = binding.make_plan(p, binding.PLAN_SYMMETRIC_DATA | binding.PLAN_BREAK_SYMMETRY | binding.PLAN_DEGREE_FILTER)
    binding.host_lib().freePattern(p)
    out, inn = planref.csr_lists(n, s, d)
    canon = planref.run_plan(plan, n, out, inn, labels=dl, symmetric=True)
    assert len(canon) * plan.n_aut == cnt
    got = sorted(tuple(x + 1 for x in r) for r in planref.expand_automorphisms(plan, canon))
    assert got == sorted(map(tuple, rows.tolist()))


# ---- the program's command line (reference: checkArgs, src/gpm-par.c:49-69) ----

EXE = os.path.join(util.ROOT, "gpm-par.exe")


def run_exe(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([EXE] + args, capture_output=True, text=True, env=e, timeout=120)


def test_cli_wrong_argc():
    r = run_exe([])
    assert r.returncode == 1
    assert r.stderr == "ERROR: files not specified\n"
    r = run_exe(["a", "b", "c"])
    assert r.returncode == 1 and r.stderr == "ERROR: files not specified\n"


def test_cli_missing_input(tmp_path):
    r = run_exe([str(tmp_path / "nope.txt"), str(tmp_path / "out.txt")])
    assert r.returncode == 1
    assert r.stderr == f"ERROR: file not found {tmp_path / 'nope.txt'}\n"


def test_cli_nonzero_mpi_ranks_exit_quietly(tmp_path):
    r = run_exe([], env={"PMI_RANK": "1"})
    assert r.returncode == 0 and r.stderr == "" and r.stdout == ""


def test_text_after_the_pattern_is_ignored_unless_it_is_a_labels_trailer(tmp_path, monkeypatch):
    """The reference stops reading at the blank line that ends the pattern.  Only a line holding the
    single word `labels` starts the trailer; GPM_LABELS=0 ignores even that."""
    base = "3\n1 2\n2\n3\n2 1\n3\n\n2\n1 1\n2\n\n"
    for tail, labelled in (("labelsfoo 1 2\n", False), ("labels_are_nice\n", False), ("7 7\n", False),
                           ("labels\n0\n1\n0\n0\n1\n", True)):
        p = tmp_path / "x.in"
        p.write_text(base + tail)
        data, pat = binding.readGraph(str(p))
        assert (data["labels"] is not None) == labelled, tail
        binding.host_lib().freePattern(pat)
    monkeypatch.setenv("GPM_LABELS", "0")
    data, pat = binding.readGraph(str(p))
    assert data["labels"] is None
    binding.host_lib().freePattern(pat)
    monkeypatch.delenv("GPM_LABELS")
    p.write_text(base + "labels\n0\n1\n")  # too few labels
    with pytest.raises(binding.GpmError):
        binding.readGraph(str(p))


def test_host_code_and_oracle_under_sanitizers(tmp_path):
    """`make sanitize`: parser, threaded loader, CSR cache, planner (host_check) and the oracle built with
    AddressSanitizer + UndefinedBehaviorSanitizer run over the hand-written cases, generated inputs and
    malformed inputs: no sanitizer report, the same digests as the plain build, the golden outputs."""
    import subprocess
    subprocess.run(["make", "-C", util.ROOT, "sanitize"], check=True, stdout=subprocess.DEVNULL)
    san = os.path.join(util.ROOT, "mpi_amd", "lib", "san")
    files = [util.small_paths(n)[0] for n in util.small_cases()]
    for name in ("rand40_dir_tree5_dir", "rmat8_house", "plaw9_tree5_labelled4", "rmat10_clique4", "rand9_sym_clique4",
                 "skew8_k6_minus_e"):
        spec = next(g for g in util.golden_specs() if g["name"] == name)
        f = tmp_path / (name + ".in")
        f.write_text(util.build_input(spec)[0])
        files.append(str(f))
    body = "".join(f"{v} 2\n{v % 50 + 1}\n{(v + 7) % 50 + 1}\n" for v in range(1, 51))
    odd = {
        "crlf": ("50\n" + body + "\n2\n1 1\n2\n").replace("\n", "\r\n"),
        "degree_too_large": "50\n" + body.replace("3 2\n", "3 3\n", 1) + "\n2\n1 1\n2\n",
        "id_out_of_range": "50\n" + body.replace("\n9\n", "\n99\n", 1) + "\n2\n1 1\n2\n",
        "no_blank_line": "50\n" + body,
        "long_line": "50\n" + body.replace("\n9\n", "\n" + "0" * 70 + "9\n", 1) + "\n2\n1 1\n2\n",
        "truncated": "3\n1 2\n2\n",
        "empty": "",
        "pattern_too_large": "3\n1 1\n2\n\n11\n1 1\n2\n",
        "bad_labels": "3\n1 1\n2\n\n2\n1 1\n2\n\nlabels\n1\n",
        "disconnected_pattern": "3\n1 1\n2\n\n3\n1 1\n2\n",
    }
    for name, text in odd.items():
        f = tmp_path / (name + ".in")
        f.write_bytes(text.encode())
        files.append(str(f))
    env = dict(os.environ)
    env.update({"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1",
                "GPM_LOAD_MIN_BYTES": "0"})
    for threads, cache in (("1", "0"), ("5", "0"), ("5", "1"), ("5", "1")):  # the last run loads the side-cars
        env.update({"GPM_LOAD_THREADS": threads, "GPM_CSR_CACHE": cache, "GPM_CSR_CACHE_DIR": str(tmp_path)})
        a = subprocess.run([os.path.join(san, "host_check")] + files, capture_output=True, text=True, env=env, timeout=600)
        b = subprocess.run([os.path.join(san, "host_check_plain")] + files, capture_output=True, text=True, env=env,
                           timeout=600)
        assert a.returncode == 0 and a.stderr == "", a.stderr[-3000:]
        assert a.stdout == b.stdout and a.stdout.count("\n") == len(files)
    for name in util.small_cases():
        inp, out = util.small_paths(name)
        got = tmp_path / "o.txt"
        r = subprocess.run([os.path.join(san, "gpm_oracle"), inp, str(got)], capture_output=True, text=True, env=env,
                           timeout=600)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        assert got.read_bytes() == open(out, "rb").read()


def _connected_orders(pattern):
    import itertools
    P, arcs = pattern
    nbr = {v: set() for v in range(1, P + 1)}
    for a, b in arcs:
        if a != b:
            nbr[a].add(b)
            nbr[b].add(a)
    return [perm for perm in itertools.permutations(range(1, P + 1))
            if all(any(w in nbr[perm[k]] for w in perm[:k]) for k in range(1, P))]


def test_binding_order_chosen_by_cost():
    """With the degree distribution of a skewed graph the planner moves the house pattern's third vertex to the
    lower end of its ordering bound and keeps the closed-form tail; patterns whose default order is already
    the cheapest of its class (octahedron, cliques, trees) keep it; the reference order and explicit orders
    are still available."""
    off, adj = gen.rmat(13, 16, seed=1)
    o, _ = gen.csr_to_numpy(off, adj)
    st = binding.graph_stats(1 << 13, o)
    flags = binding.PLAN_SYMMETRIC_DATA | binding.PLAN_BREAK_SYMMETRY | binding.PLAN_DEGREE_FILTER

    def order_of(name, stats, extra=0):
        p = binding.make_pattern(gen.PATTERNS[name])
        try:
            plan = binding.make_plan(p, flags | extra, stats)
        finally:
            binding.host_lib().freePattern(p)
        return [x + 1 for x in list(plan.order)[:plan.n_levels]], plan

    house, plan = order_of("house", st)
    assert house != order_of("house", None)[0]
    assert house[:2] in ([1, 4], [4, 1], [2, 3], [3, 2], [1, 2], [2, 1]) and sorted(house[-2:]) in ([3, 5], [4, 5])
    # the third vertex hangs off the vertex the bounds make the smaller one
    lower = {plan.lower_level[k][j] for k in range(5) for j in range(plan.n_lower[k])}
    assert plan.src_level[2] in lower or plan.n_lower[plan.src_level[2]] == 0
    for name in ("k6_minus_pm", "clique4", "clique5", "tree5", "diamond", "k6_minus_e"):
        assert order_of(name, st)[0] == order_of(name, None)[0], name
    assert order_of("house", st, binding.PLAN_REFERENCE_ORDER)[0] == [1, 2, 3, 4, 5]


def test_explicit_binding_orders_are_valid_plans(monkeypatch):
    """GPM_PLAN_ORDER="3,1,2,...": every connected order gives a plan whose levels list exactly the pattern's
    edges (tree arc + constraints); an order with a disconnected prefix is rejected."""
    pat = gen.PATTERNS["house"]
    P, arcs = pat
    edges = {frozenset((a, b)) for a, b in arcs}
    for perm in _connected_orders(pat):
        monkeypatch.setenv("GPM_PLAN_ORDER", ",".join(map(str, perm)))
        p = binding.make_pattern(pat)
        plan = binding.make_plan(p, binding.PLAN_SYMMETRIC_DATA)
        binding.host_lib().freePattern(p)
        assert [x + 1 for x in list(plan.order)[:P]] == list(perm)
        seen = set()
        for k in range(1, P):
            seen.add(frozenset((perm[k], perm[plan.src_level[k]])))
            for j in range(plan.n_cons[k]):
                seen.add(frozenset((perm[k], perm[plan.cons_level[k][j]])))
        assert seen == edges, perm
    monkeypatch.setenv("GPM_PLAN_ORDER", "1,3,2,4,5")  # 3 is not adjacent to 1
    p = binding.make_pattern(pat)
    with pytest.raises(binding.GpmError):
        binding.make_plan(p, binding.PLAN_SYMMETRIC_DATA)
    binding.host_lib().freePattern(p)
