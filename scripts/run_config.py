"""Run one of the five benchmark configurations at full size on one GPU (count-only)
and print the count, the wall time and the device statistics."""
import json, sys, time
import torch
sys.path.insert(0, '.')
from mpi_amd import binding, gen

cfg = int(sys.argv[1])
scale_override = int(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device('cuda')
labels = plabels = None
if cfg == 1:
    off, adj = gen.erdos_renyi(10000, 50000, seed=1, device=dev); n = 10000; pat = 'triangle'
elif cfg == 2:
    s = scale_override or 20
    off, adj = gen.rmat(s, 16, seed=1, device=dev); n = 1 << s; pat = 'cycle4'
elif cfg == 3:
    s = scale_override or 24
    off, adj = gen.rmat(s, 16, seed=1, device=dev); n = 1 << s; pat = 'clique4'
elif cfg == 4:
    s = scale_override or 22
    n = 1 << s
    off, adj = gen.power_law(n, avg_deg=16, exponent=2.1, seed=1, device=dev); pat = 'tree5'
    labels = gen.vertex_labels(n, 16, seed=1, device=dev).cpu().numpy(); plabels = [0, 1, 2, 3, 1]
elif cfg == 5:
    s = scale_override or 18
    off, adj = gen.rmat(s, 8, abcd=(0.7, 0.1, 0.1, 0.1), seed=1, device=dev); n = 1 << s; pat = 'k6_minus_pm'
off_h, adj_h = off.cpu().numpy(), adj.cpu().numpy()
del off, adj
torch.cuda.empty_cache()
m = binding.Matcher(1)
info = m.upload(n, off_h, adj_h, labels=labels)
plan = m.set_pattern(gen.PATTERNS[pat], labels=plabels)
t0 = time.time()
count = m.findMatches(count_only=True)
dt = time.time() - t0
st = m.stats()
print(json.dumps({"config": cfg, "pattern": pat, "n": n, "arcs": int(off_h[-1]), "max_deg": info.max_out_deg,
                  "n_aut": plan.n_aut, "matches": count, "seconds": round(dt, 3),
                  "matches_per_s": count / dt if dt > 0 else None, "alg_bytes": st.alg_bytes,
                  "rows_in": list(st.rows_in)[:plan.n_levels], "tail_ms": st.ms_tail_kernel,
                  "extend_ms": st.ms_extend_kernel}))
