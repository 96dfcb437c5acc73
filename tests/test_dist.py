"""The N>1 path on CPU: two gloo ranks share one chunk cursor, and counts /
match lists are combined by all-reduce / all-gather.  The per-chunk search is
played by the oracle here (test stand-in for the GPU), so what is under test
is the sharding protocol itself."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tag, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mpi_amd import binding, dist as gdist, gen
    from tests import util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, s, d = util.build_graph({"graph": "rmat", "scale": 8, "ef": 8, "seed": 1})
    pat = gen.PATTERNS["clique4"]
    host = binding.host_lib()
    host.claimChunk.restype = ctypes.c_uint64
    host.claimChunk.argtypes = [ctypes.c_void_p]
    chunk = 16
    claimed = []

    def search_chunks(cursor_address, count_only):
        rows, count = [], 0
        while True:
            c = host.claimChunk(cursor_address)
            lo = c * chunk
            if lo >= n:
                break
            claimed.append(c)
            cnt, r, _ = util.oracle_matches(n, s, d, pat, root_lo=lo + 1, root_hi=min(n, lo + chunk))
            count += cnt
            rows.append(r)
        return count if count_only else (np.concatenate(rows) if rows else np.zeros((0, pat[0]), np.int32))

    cur = gdist.open_cursor(tag)
    total = gdist.sharded_search(search_chunks, cur, count_only=True)
    first = list(claimed)
    claimed.clear()
    rows = gdist.sharded_search(search_chunks, cur, count_only=False)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), rows)
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
        f.write(f"{total}\n{' '.join(map(str, first))}\n")
    dist.barrier()
    cur.close()
    dist.destroy_process_group()


def test_two_ranks_share_the_work_and_agree(tmp_path):
    from mpi_amd import gen
    from tests import util
    world = 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, f"t{os.getpid()}", str(tmp_path)), nprocs=world, join=True)
    n, s, d = util.build_graph({"graph": "rmat", "scale": 8, "ef": 8, "seed": 1})
    want, want_rows, _ = util.oracle_matches(n, s, d, gen.PATTERNS["clique4"])
    seen = []
    for r in range(world):
        total, chunks = open(tmp_path / f"r{r}.txt").read().splitlines()[:2] + [""] * 0
        assert int(total) == want
        seen += [int(x) for x in chunks.split()]
        rows = np.load(tmp_path / f"r{r}.npy")
        assert sorted(map(tuple, rows.tolist())) == sorted(map(tuple, want_rows.tolist()))
    assert sorted(seen) == list(range((n + 15) // 16)), "every chunk is taken exactly once"
