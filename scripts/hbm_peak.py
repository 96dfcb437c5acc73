"""HBM bandwidth of this GPU two ways: from the device properties (memory clock x bus width) and
measured (device-to-device copy and a read-only sum over 8 GiB)."""
import json, time, torch
p = torch.cuda.get_device_properties(0)
out = {"name": p.name, "total_memory_GiB": round(p.total_memory / 2**30, 1), "compute_units": p.multi_processor_count}
clk_khz = getattr(p, "memory_clock_rate", None)
bus = getattr(p, "memory_bus_width", None)
if clk_khz and bus:
    out["memory_clock_MHz"] = clk_khz / 1e3
    out["bus_width_bits"] = bus
    out["peak_GBps_from_properties_ddr"] = 2 * clk_khz * 1e3 * bus / 8 / 1e9
n = 1 << 30  # 8 GiB of int64
a = torch.empty(n, dtype=torch.int64, device="cuda").fill_(1)
b = torch.empty_like(a)
for name, fn, bytes_moved in (("copy_GBps", lambda: b.copy_(a), 16 * n), ("read_GBps", lambda: a.sum(), 8 * n)):
    fn(); torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = max(best, bytes_moved / dt / 1e9)
    out[name] = round(best, 1)
print(json.dumps(out, indent=1))
