"""End to end through the drop-in program: generate one of the benchmark graphs, write it in the
reference's text format, run ./gpm-par.exe on the file (count-only) and report its two stdout
lines, the count and the wall time, next to the size of the input file."""
import json, os, subprocess, sys, tempfile, time
import torch
sys.path.insert(0, '.')
from mpi_amd import binding, gen

cfg = int(sys.argv[1])
scale = int(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device('cuda')
labels = plabels = None
if cfg == 2:
    s = scale or 20; off, adj = gen.rmat(s, 16, seed=1, device=dev); n = 1 << s; pat = 'cycle4'
elif cfg == 3:
    s = scale or 24; off, adj = gen.rmat(s, 16, seed=1, device=dev); n = 1 << s; pat = 'clique4'
elif cfg == 4:
    s = scale or 22; n = 1 << s
    off, adj = gen.power_law(n, avg_deg=16, exponent=2.1, seed=1, device=dev); pat = 'tree5'
    labels = gen.vertex_labels(n, 16, seed=1, device=dev).cpu().numpy(); plabels = [0, 1, 2, 3, 1]
else:
    raise SystemExit("configs 2, 3, 4")
off_h, adj_h = off.cpu().numpy(), adj.cpu().numpy()
del off, adj
torch.cuda.empty_cache()
tmp = tempfile.mkdtemp(prefix="gpm_", dir=os.environ.get("GPM_TMP", "/tmp"))
inp, out = os.path.join(tmp, "in.txt"), os.path.join(tmp, "out.txt")
t0 = time.time()
binding.write_input_csr(inp, n, off_h, adj_h, gen.PATTERNS[pat], labels, plabels)
t_write = time.time() - t0
env = dict(os.environ, GPM_COUNT_ONLY="1", GPM_STATS="1", GPM_LOAD_DEBUG="1")
walls = []
for _ in range(3 if env.get("GPM_CSR_CACHE") == "1" else 1):  # with the side-car cache: store, then load twice
    t0 = time.time()
    r = subprocess.run(["./gpm-par.exe", inp, out], capture_output=True, text=True, env=env)
    walls.append(round(time.time() - t0, 2))
wall = walls[-1]
size = os.path.getsize(inp)
cache_bytes = os.path.getsize(inp + ".gpmcsr") if os.path.exists(inp + ".gpmcsr") else 0
for f in (inp, out, inp + ".gpmcsr"):
    if os.path.exists(f):
        os.remove(f)
os.rmdir(tmp)
print(json.dumps({"config": cfg, "pattern": pat, "n": n, "arcs": int(off_h[-1]), "input_bytes": size,
                  "write_input_s": round(t_write, 2), "program_wall_s": round(wall, 2), "all_walls_s": walls, "cache_bytes": cache_bytes, "rc": r.returncode,
                  "stdout": r.stdout.splitlines(), "stderr": r.stderr.splitlines()[-12:]}))
