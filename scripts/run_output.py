"""Output mode end to end: ./gpm-par.exe writes every match of a pattern on an R-MAT graph; reports
lines written, bytes, the program's wall time and the md5 of the sorted lines when small."""
import json, os, subprocess, sys, tempfile, time
import torch
sys.path.insert(0, '.')
from mpi_amd import binding, gen

scale, ef, pat = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device('cuda')
off, adj = gen.rmat(scale, ef, seed=1, device=dev)
n = 1 << scale
off_h, adj_h = off.cpu().numpy(), adj.cpu().numpy()
del off, adj
tmp = tempfile.mkdtemp(prefix="gpm_", dir=os.environ.get("GPM_TMP", "/tmp"))
inp, out = os.path.join(tmp, "in.txt"), os.path.join(tmp, "out.txt")
binding.write_input_csr(inp, n, off_h, adj_h, gen.PATTERNS[pat])
env = dict(os.environ, GPM_STATS="1")
t0 = time.time()
r = subprocess.run(["./gpm-par.exe", inp, out], capture_output=True, text=True, env=env)
wall = time.time() - t0
size = os.path.getsize(out)
with open(out, "rb") as f:
    lines = sum(chunk.count(b"\n") for chunk in iter(lambda: f.read(1 << 24), b""))
os.remove(inp); os.remove(out); os.rmdir(tmp)
print(json.dumps({"scale": scale, "ef": ef, "pattern": pat, "arcs": int(off_h[-1]), "lines": lines,
                  "output_bytes": size, "program_wall_s": round(wall, 2), "rc": r.returncode,
                  "stdout": r.stdout.splitlines(), "stderr": r.stderr.splitlines()[-2:]}))
