"""Tree patterns whose leaves hang off up to three inner vertices, with the leaf-suffix closed form on and off."""
import os, sys, time, torch
sys.path.insert(0, '.')
from mpi_amd import binding, gen

def und(p):
    return sorted({(a, b) for a, b in p} | {(b, a) for a, b in p})

PATS = {"caterpillar6": (6, und([(1, 2), (2, 3), (1, 4), (2, 5), (3, 6)])),
        "spider7": (7, und([(1, 2), (1, 3), (1, 4), (2, 5), (3, 6), (4, 7)])),
        "cat_221": (8, und([(1, 2), (2, 3), (1, 4), (1, 5), (2, 6), (2, 7), (3, 8)]))}
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 14
off, adj = gen.rmat(scale, 16, seed=1, device=torch.device('cuda'))
for name, pat in PATS.items():
    for sw in ("0", "1"):
        os.environ["GPM_NO_LEAF_SUFFIX"] = sw
        m = binding.Matcher(1)
        m.upload(1 << scale, off.cpu().numpy(), adj.cpu().numpy())
        m.set_pattern(pat)
        t = time.time(); c = m.findMatches(count_only=True); dt = time.time() - t
        print(f"{name:14s} closed form {'off' if sw == '1' else 'on '} count {c:24d} seconds {dt:8.3f}", flush=True)
        m.close()
