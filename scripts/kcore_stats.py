"""Experiment: degeneracy (k-core) ordering vs degree ordering on the benchmark graph."""
import sys, time, torch
sys.path.insert(0, '.')
from mpi_amd import gen
scale = int(sys.argv[1])
dev = torch.device('cuda')
off, adj = gen.rmat(scale, 16, seed=1, device=dev)
n = 1 << scale
deg = (off[1:] - off[:-1]).clone()
src = torch.repeat_interleave(torch.arange(n, device=dev), deg)
dst = adj.long()
alive = torch.ones(n, dtype=torch.bool, device=dev)
order = torch.full((n,), -1, dtype=torch.int64, device=dev)   # removal round
cur = deg.clone()
k = 0
pos = 0
rounds = 0
t0 = time.time()
while pos < n:
    rem = alive & (cur <= k)
    nr = int(rem.sum())
    if nr == 0:
        k = int(cur[alive].min())
        continue
    rounds += 1
    idx = torch.nonzero(rem).squeeze(1)
    order[idx] = pos + torch.arange(nr, device=dev)   # ties inside a round by id
    pos += nr
    alive[idx] = False
    # decrement neighbours of removed vertices
    m = rem[src] & alive[dst]
    cur -= torch.bincount(dst[m], minlength=n)
print('degeneracy', k, 'rounds', rounds, 'time', time.time() - t0)
up = order[dst] > order[src]
dplus = torch.bincount(src[up], minlength=n)
print('max d+ (core order)', int(dplus.max()), 'sum d+^2', float((dplus.double() ** 2).sum()))
rank = torch.empty(n, dtype=torch.int64, device=dev)
rank[torch.argsort(deg * n + torch.arange(n, device=dev))] = torch.arange(n, device=dev)
up2 = rank[dst] > rank[src]
dp2 = torch.bincount(src[up2], minlength=n)
print('max d+ (degree order)', int(dp2.max()), 'sum d+^2', float((dp2.double() ** 2).sum()))
for thr in [256, 512, 1024]:
    print('d+ >', thr, 'core', int((dplus > thr).sum()), 'degree', int((dp2 > thr).sum()))
