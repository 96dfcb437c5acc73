"""Streaming-read rate of this GPU as a plain library reduction reaches it (calibrates what 'HBM-bound' can mean here)."""
import torch, time
x = torch.ones(1 << 31, dtype=torch.int32, device="cuda")  # 8 GiB
for name, fn in (("sum", lambda: x.sum()), ("max", lambda: x.max())):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(name, "GB/s", round(x.numel() * 4 / dt / 1e9, 1))
y = torch.empty_like(x)
y.copy_(x); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): y.copy_(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print("copy GB/s (read+write)", round(2 * x.numel() * 4 / dt / 1e9, 1))
# a 256 MiB working set read over and over: the rate of the last-level cache in front of HBM
z = x[: 1 << 25]
z.sum(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50): z.sum()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 50
print("128 MiB resident sum GB/s", round(z.numel() * 4 / dt / 1e9, 1))
