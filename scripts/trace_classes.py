"""Per-launch times of one benchmark configuration, summed by kernel family and size class.

    python scripts/trace_classes.py <config> [inflight]

Runs one warm-up search and one traced search (GPM_TRACE=1 makes the library print every launch's HIP-event
time on stderr) in a child process and adds the lines up.  inflight = 1 runs the clique path with one chunk
in flight, so launches do not overlap and family times add up to the wall time."""
import collections
import os
import re
import subprocess
import sys

cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
inflight = sys.argv[2] if len(sys.argv) > 2 else None
env = dict(os.environ, GPM_TRACE="1")
if inflight:
    env["GPM_CLIQUE_INFLIGHT"] = inflight
p = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                    "--no-extra-search"], env=env, capture_output=True, text=True)
print(p.stdout.strip()[:400])
ms = collections.defaultdict(float)
launches = collections.Counter()
rows = collections.Counter()
lines = [l for l in p.stderr.splitlines() if l.startswith("gpm trace: k_")]
# the warm-up search printed the same number of lines before: keep the second half
half = lines[len(lines) // 2:]
for l in half:
    m = re.match(r"gpm trace: (\S+) class (\d+) rows (\d+) ms ([0-9.]+)", l)
    if not m:
        continue
    key = (m.group(1), int(m.group(2)))
    ms[key] += float(m.group(4))
    launches[key] += 1
    rows[key] += int(m.group(3))
print(f"config {cfg} inflight {inflight or 'default'}")
for key in sorted(ms):
    print(f"{key[0]:14s} class {key[1]}  launches {launches[key]:4d}  rows {rows[key]:10d}  ms {ms[key]:9.3f}")
print("sum ms", round(sum(ms.values()), 3))
