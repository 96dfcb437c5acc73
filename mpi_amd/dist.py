"""Multi-GPU sharding of the match search: one process per GPU.

The data graph is replicated on every GPU; the roots are cut into chunks
that all ranks of the node take from one shared cursor (work stealing), so no
data-path collective runs during the search.  The only exchange is the result
gathering the reference does with one MPI message per match (reference:
src/gpm-par.c:181 and :608-621): here a single all-reduce of the counts and,
in output mode, an all-gather of the match lists (RCCL over xGMI when the
process group is "nccl", gloo in the CPU tests).
"""
from __future__ import annotations

import ctypes
import mmap
import os

import numpy as np
import torch
import torch.distributed as dist


class SharedCursor:
    """Two zero-initialised 64-bit words (a cache line apart) in /dev/shm that every rank of the node maps.
    `address` is the first word; searches that follow one another closely alternate between the two
    (`begin_step`)."""

    def __init__(self, name: str, create: bool):
        self.path = os.path.join("/dev/shm", name)
        if create:
            with open(self.path, "wb") as f:
                f.write(b"\0" * 128)
        self._f = open(self.path, "r+b")
        self._map = mmap.mmap(self._f.fileno(), 128)
        self._word = ctypes.c_uint64.from_buffer(self._map)
        self._word1 = ctypes.c_uint64.from_buffer(self._map, 64)
        self.address = ctypes.addressof(self._word)
        self.addresses = (self.address, ctypes.addressof(self._word1))
        self.epoch = 0
        self._owner = create

    def reset(self):
        self._word.value = 0

    def zero(self, which: int):
        (self._word1 if which else self._word).value = 0

    @property
    def value(self):
        return self._word.value

    def close(self):
        del self._word, self._word1
        self._map.close()
        self._f.close()
        if self._owner and os.path.exists(self.path):
            os.unlink(self.path)


def open_cursor(tag: str) -> SharedCursor:
    """Rank 0 creates the cursor file, everybody maps it (collective)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    name = f"gpm_cursor_{tag}"
    if rank == 0:
        cur = SharedCursor(name, create=True)
    if dist.is_initialized():
        dist.barrier()
    if rank != 0:
        cur = SharedCursor(name, create=False)
    if dist.is_initialized():
        dist.barrier()
    return cur


def reset_cursor(cur: SharedCursor):
    """Collective: every rank has left the previous search, then rank 0 zeroes the cursor."""
    if not dist.is_initialized():
        cur.reset()
        return
    dist.barrier()
    if dist.get_rank() == 0:
        cur.reset()
    dist.barrier()


def begin_step(cur: SharedCursor) -> int:
    """The address of the cursor word for the next search of a series in which EVERY search ends with a
    collective all ranks join (the all-reduce of the counts).  No barrier is needed: search s takes its
    chunks from word s mod 2, and rank 0 zeroes the other word before it starts search s.  That word last
    served search s-1, which every rank had left before the collective that ended it completed; and no
    rank starts search s+1 before the collective that ends search s, which rank 0 joins after the zeroing.
    The first search finds word 0 zero as `open_cursor` left it."""
    here = cur.epoch & 1
    if not dist.is_initialized() or dist.get_rank() == 0:
        cur.zero(here ^ 1)
    cur.epoch += 1
    return cur.addresses[here]


def all_reduce_count(local_count: int, device="cpu") -> int:
    """Sum of the per-rank embedding counts (all-reduce, 64-bit)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return int(local_count)
    t = torch.tensor([local_count], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def all_gather_matches(rows: np.ndarray, device="cpu") -> np.ndarray:
    """All ranks' match rows on every rank: all-gather of the row counts, then of
    the lists padded to the longest one."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    P = rows.shape[1]
    n_local = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    longest = max(max(counts), 1)
    padded = torch.zeros((longest, P), dtype=torch.int32, device=device)
    padded[:rows.shape[0]] = torch.from_numpy(np.ascontiguousarray(rows)).to(device)
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)])


def sharded_search(search_chunks, cursor: SharedCursor, count_only: bool, device="cpu"):
    """Run `search_chunks(cursor_address, count_only)` on every rank -- it takes
    root chunks from the shared cursor until none are left and returns its local
    count (count_only) or its local rows -- and combine the results."""
    reset_cursor(cursor)
    local = search_chunks(cursor.address, count_only)
    if count_only:
        return all_reduce_count(local, device)
    return all_gather_matches(local, device)


def timed_steps(step, steps: int, warmup: int, fence, device="cpu"):
    """The benchmark's measuring protocol, shared by every rank: `warmup` untimed steps, then exactly
    `steps` steps bracketed by `fence()` (barrier + device synchronisation) on both sides.  Returns
    (seconds -- the maximum over ranks --, the last step's result, per-rank lists gathered on every rank of
    what `step` reported as busy milliseconds and chunks taken).  `step()` returns (result, busy_ms, chunks)."""
    import time
    result = None
    for _ in range(warmup):
        result, _, _ = step()
    fence()
    t0 = time.perf_counter()
    busy_ms, chunks = 0.0, 0
    for _ in range(steps):
        result, b, c = step()
        busy_ms += b
        chunks += c
    fence()
    elapsed = time.perf_counter() - t0
    busy, taken = [busy_ms], [chunks]
    if dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([busy_ms, float(chunks)], dtype=torch.float64, device=device)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        busy = [float(p[0].item()) for p in parts]
        taken = [int(p[1].item()) for p in parts]
    return elapsed, result, busy, taken
