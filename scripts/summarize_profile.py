#!/usr/bin/env python3
"""Condense a scripts/profile.sh run (gpurun_out/prof_<tag>/) into the files kept
under profiles/: the rocprofv3 per-kernel time table, the counter sums of the
search kernels, and the HBM bytes per launch that bench.py reports as
roofline.traffic.

Counter corrections (MI355X, measured by scripts/calib_fetch under the same
tool): FETCH_SIZE is in KiB and reads exactly half of the bytes of coalesced 4-
and 16-byte-per-lane streams, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE
is in KiB and exact for coalesced stores."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join("gpurun_out", f"prof_{tag}")
dst = "profiles"
os.makedirs(dst, exist_ok=True)
SEARCH = ("k_clique", "k_clq_", "k_cycle4", "k_octa", "k_tail", "k_leaf_pair", "k_leaf_suffix", "k_extend", "k_ranges", "k_filter_roots", "k_batch_end",
          "k_sum_slots")
# kernel family (the names gpm_hip_kernel_name returns) -> prefixes of the kernels in it
FAMILIES = {"k_tail": ("k_tail", "k_leaf_pair", "k_leaf_suffix"), "k_extend": ("k_extend",), "k_clique": ("k_clique<", "k_clique_tiny<"),
            "k_clq_build": ("k_clq_build",), "k_clq_count4": ("k_clq_count4",), "k_cycle4": ("k_cycle4_light", "k_cycle4_heavy", "k_cycle4_ranges"), "k_octa": ("k_octa",)}


def fingerprint():
    """Hash of the kernel sources the counters were collected from; bench.py reports the traffic only
    while the sources still hash to it."""
    import hashlib
    h = hashlib.sha256()
    for name in ("gpm_kernels.hpp", "gpm_device.hpp", "gpm_hip.hip"):
        with open(os.path.join("mpi_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    with open(os.path.join(src, sub, "t_counter_collection.csv")) as f:
        seen = set()
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (r["Dispatch_Id"], r["Counter_Name"]) not in seen and r["Counter_Name"] == list(agg[k])[0]:
                calls[k] += 1
            seen.add((r["Dispatch_Id"], r["Counter_Name"]))
    return agg, calls


# 1. per-kernel time table as rocprofv3 printed it (top rows + every search kernel)
rows = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
keep = [r for i, r in enumerate(rows) if i < 12 or short(r["Name"]).startswith(SEARCH)]
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in keep:
        r = dict(r)
        r["Name"] = short(r["Name"])
        w.writerow(r)

# 1b. the same table from the pass with one chunk in flight (launches do not overlap), when it was made
alone = {}
alone_path = os.path.join(src, "trace_alone", "t_kernel_stats.csv")
if os.path.exists(alone_path):
    arows = list(csv.DictReader(open(alone_path)))
    alone = {short(r["Name"]): r for r in arows}
    with open(os.path.join(dst, f"{tag}_kernel_stats_one_chunk_in_flight.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(arows[0].keys()))
        w.writeheader()
        for i, r in enumerate(arows):
            if i < 12 or short(r["Name"]).startswith(SEARCH):
                r = dict(r)
                r["Name"] = short(r["Name"])
                w.writerow(r)

# 2. counters of the search kernels
fetch, fcalls = counters("fetch")
write, _ = counters("write")
sq, _ = counters("sq")
trace = {short(r["Name"]): r for r in rows}
summary = {}
for k in sorted(set(fetch) | set(sq)):
    if not k.startswith(SEARCH):
        continue
    t = trace.get(k, {})
    calls = int(t.get("Calls", fcalls[k]) or 0)
    rd = 2.0 * fetch[k].get("FETCH_SIZE", 0.0) * 1024.0
    wr = write[k].get("WRITE_SIZE", 0.0) * 1024.0
    s = sq[k]
    summary[k] = {
        "calls": calls,
        "total_ms": float(t.get("TotalDurationNs", 0)) / 1e6,
        "avg_ms": float(t.get("AverageNs", 0)) / 1e6,
        "hbm_read_bytes": rd, "hbm_write_bytes": wr,
        "FETCH_SIZE_KiB_raw": fetch[k].get("FETCH_SIZE", 0.0), "WRITE_SIZE_KiB_raw": write[k].get("WRITE_SIZE", 0.0),
        "SQ_LDS_BANK_CONFLICT": s.get("SQ_LDS_BANK_CONFLICT", 0.0), "SQ_LDS_IDX_ACTIVE": s.get("SQ_LDS_IDX_ACTIVE", 0.0),
        "lds_conflict_share": (s.get("SQ_LDS_BANK_CONFLICT", 0.0) / s["SQ_LDS_IDX_ACTIVE"]) if s.get("SQ_LDS_IDX_ACTIVE") else None,
        "SQ_INSTS_LDS": s.get("SQ_INSTS_LDS", 0.0), "SQ_INSTS_VALU": s.get("SQ_INSTS_VALU", 0.0),
        "SQ_WAVES": s.get("SQ_WAVES", 0.0), "SQ_WAVE_CYCLES": s.get("SQ_WAVE_CYCLES", 0.0),
        "SQ_BUSY_CYCLES": s.get("SQ_BUSY_CYCLES", 0.0), "SQ_WAIT_INST_LDS": s.get("SQ_WAIT_INST_LDS", 0.0),
    }
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
    f.write("\n")

# 3. HBM bytes per launch of every kernel family, the same "per launch" basis as
#    bench.py's roofline.achieved; the counters cover every search of the profiled run
traffic = {"note": "read bytes = 2 x FETCH_SIZE KiB (gfx950 correction, calibrated with scripts/calib_fetch), "
                   "write bytes = WRITE_SIZE KiB; sums over the whole profiled run (warm-up and timed steps)",
           "source": f"profiles/{tag}_pmc_summary.json", "kernels_fingerprint": fingerprint()}
for fam, prefixes in FAMILIES.items():
    part = [v for k, v in summary.items() if k.startswith(prefixes)]
    if not part:
        continue
    calls = sum(v["calls"] for v in part)
    if not calls:
        continue
    hbm = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in part)
    ms = sum(v["total_ms"] for v in part)
    valu = sum(v["SQ_INSTS_VALU"] for v in part)
    conflict, active = sum(v["SQ_LDS_BANK_CONFLICT"] for v in part), sum(v["SQ_LDS_IDX_ACTIVE"] for v in part)
    traffic[fam] = {"launches": calls, "hbm_bytes_total": hbm, "hbm_bytes_per_launch": hbm / calls,
                    "kernel_ms_total": ms, "kernel_avg_ms": ms / calls,
                    "hbm_GBps_while_running": hbm / 1e9 / (ms / 1e3) if ms else None,
                    # wave-level vector-ALU instructions (SQ_INSTS_VALU): against 256 CUs x 1 per clock x 2.4 GHz
                    # over a launch's time this is the share of the issue rate the family uses
                    "valu_instructions_total": valu, "valu_instructions_per_launch": valu / calls,
                    "lds_conflict_share": conflict / active if active else None}
    apart = [v for k, v in alone.items() if k.startswith(prefixes)]
    acalls = sum(int(v["Calls"]) for v in apart)
    if acalls:  # launch time without overlap, and the shares of the two peaks that follow from it
        a_ms = sum(float(v["TotalDurationNs"]) for v in apart) / 1e6
        traffic[fam]["alone_launches"] = acalls
        traffic[fam]["alone_avg_ms"] = a_ms / acalls
        traffic[fam]["valu_frac_alone"] = (valu / calls) / (256 * 2.4e9 * a_ms / acalls / 1e3)
        traffic[fam]["hbm_frac_alone"] = (hbm / calls) / (8e12 * a_ms / acalls / 1e3)
# wave-level vector-ALU instructions of all search kernels per search (the profiled command runs 1 warm-up
# and 2 timed searches): the resource that bounds the clique search once HBM traffic is a third of the
# merge model's bytes
SEARCHES = 3
traffic["searches_profiled"] = SEARCHES
traffic["valu_wave_instructions_per_search"] = sum(v["SQ_INSTS_VALU"] for v in summary.values()) / SEARCHES
traffic["hbm_bytes_per_search"] = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in summary.values()) / SEARCHES
# bench.py reads hbm_traffic.json for the headline workload (configuration 3) and config<N>_hbm_traffic.json
# for the others: a tag that names a configuration (e.g. r2_config2) writes the latter
import re
cfg_tag = re.search(r"config(\d+)", tag)
# a tag of the form <round>_<name> (a pattern other than the configurations') keeps its traffic under that name
traffic_name = (f"config{cfg_tag.group(1)}_hbm_traffic.json" if cfg_tag
                else "hbm_traffic.json" if re.fullmatch(r"r\d+[a-z]?", tag) else f"{tag}_hbm_traffic.json")
with open(os.path.join(dst, traffic_name), "w") as f:
    json.dump(traffic, f, indent=1)
    f.write("\n")
for name in ("trace.json",):
    p = os.path.join(src, name)
    if os.path.exists(p):
        with open(p) as fi, open(os.path.join(dst, f"{tag}_bench_under_trace.json"), "w") as fo:
            fo.write(fi.read())
print(json.dumps({k: {"calls": v["calls"], "total_ms": round(v["total_ms"], 2), "GB_read": round(v["hbm_read_bytes"] / 1e9, 2),
                      "GB_written": round(v["hbm_write_bytes"] / 1e9, 3), "lds_conflict_share": v["lds_conflict_share"]}
                  for k, v in summary.items()}, indent=1))
