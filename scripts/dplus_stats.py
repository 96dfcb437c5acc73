import sys, torch, time
sys.path.insert(0, '.')
from mpi_amd import gen
scale = int(sys.argv[1])
dev = torch.device('cuda')
off, adj = gen.rmat(scale, 16, seed=1, device=dev)
n = 1 << scale
deg = off[1:] - off[:-1]
order = torch.argsort(deg * n + torch.arange(n, device=dev))  # asc degree, ties by id
rank = torch.empty_like(order); rank[order] = torch.arange(n, device=dev)
src = torch.repeat_interleave(torch.arange(n, device=dev), deg)
up = rank[adj.long()] > rank[src]
dplus = torch.bincount(src[up], minlength=n)
print('n', n, 'arcs', adj.numel(), 'maxdeg', int(deg.max()), 'max d+', int(dplus.max()))
q = torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999], device=dev, dtype=torch.float64)
s = torch.sort(dplus).values
print('d+ quantiles', [int(s[int(x * (n - 1))]) for x in q.tolist()])
for thr in [32, 64, 128, 256, 512, 1024, 2048, 4096]:
    sel = dplus >= thr
    print('d+ >=', thr, 'roots', int(sel.sum()), 'sum d+^2', float((dplus[sel].double() ** 2).sum()), 'sum d+^3/32', float((dplus[sel].double() ** 3).sum() / 32))
print('sum d+^2 all', float((dplus.double() ** 2).sum()))
