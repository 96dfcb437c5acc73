"""Count-only time of many patterns on one R-MAT graph, with the split between the kernel families."""
import sys, time, torch
sys.path.insert(0, '.')
from mpi_amd import binding, gen
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 16
names = sys.argv[2:] or ["triangle", "path3", "star4", "path4", "cycle4", "diamond", "clique4", "tree5", "house", "clique5",
                         "k6_minus_e", "k6_minus_pm", "clique6"]
dev = torch.device('cuda')
off, adj = gen.rmat(scale, 16, seed=1, device=dev)
n = 1 << scale
m = binding.Matcher(1)
m.upload(n, off.cpu().numpy(), adj.cpu().numpy())
lib = binding.hip_lib()
for pat in names:
    plan = m.set_pattern(gen.PATTERNS[pat])
    t = time.time(); c = m.findMatches(count_only=True); dt = time.time() - t
    st = m.stats()
    fam = {lib.gpm_hip_kernel_name(k).decode(): round(st.kernel_ms[k], 1) for k in range(binding.KINDS) if st.kernel_launches[k]}
    print(f"{pat:12s} aut {plan.n_aut:4d} matches {c:22d} seconds {dt:8.3f} rows {list(st.rows_in)[:plan.n_levels]} {fam}", flush=True)
m.close()
