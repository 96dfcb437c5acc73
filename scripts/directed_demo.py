"""Directed data graph (every edge of an R-MAT graph kept in one direction only, low id -> high id)
and the acyclic 4-clique pattern: each undirected 4-clique has exactly one such embedding."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from mpi_amd import binding, gen
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
dev = torch.device('cuda')
off, adj = gen.rmat(scale, 16, seed=1, device=dev)
n = 1 << scale
deg = off[1:] - off[:-1]
src = torch.repeat_interleave(torch.arange(n, device=dev), deg)
keep = adj.long() > src
s, d = src[keep], adj[keep].long()
oo = torch.zeros(n + 1, dtype=torch.int64, device=dev); oo[1:] = torch.cumsum(torch.bincount(s, minlength=n), 0)
order = torch.argsort(d * n + s)  # in-CSR: by target, sources ascending
io = torch.zeros(n + 1, dtype=torch.int64, device=dev); io[1:] = torch.cumsum(torch.bincount(d, minlength=n), 0)
m = binding.Matcher(1)
m.upload(n, oo.cpu().numpy(), d.to(torch.int32).cpu().numpy(), io.cpu().numpy(), s[order].to(torch.int32).cpu().numpy())
for pat in ("k4_dag", "tri_dir"):
    m.set_pattern(gen.PATTERNS[pat])
    t = time.time(); c = m.findMatches(count_only=True); dt = time.time() - t
    st = m.stats()
    print(pat, "arcs", int(keep.sum()), "matches", c, "seconds", round(dt, 3), "tail ms", round(st.ms_tail_kernel, 1), "extend ms", round(st.ms_extend_kernel, 1), flush=True)
m.close()
