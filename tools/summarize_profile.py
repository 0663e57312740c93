"""Copies the rocprofv3 summaries of a `scripts/gpu_round1.sh` run from gpurun_out/ into profiles/
and derives per-launch HBM traffic for the m355 kernels as MI355X_MICROARCH.md prescribes:
FETCH_SIZE and WRITE_SIZE from separate --pmc passes, in KiB; on gfx950 FETCH_SIZE counts a wide
coalesced streaming read at half its bytes, so reads = 2 * FETCH_SIZE; WRITE_SIZE is exact.
usage: python tools/summarize_profile.py r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime, reverse=True)  # gpurun_out/ accumulates: newest pass
    return hits[0] if hits else None


stats = one("prof_trace/*/*_kernel_stats.csv")
if stats:
    rows = [r for r in csv.DictReader(open(stats)) if "m355" in r["Name"] or "k_final" in r["Name"]]
    with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    shutil.copy(stats, os.path.join(dst, f"{tag}_bench_kernel_stats_all.csv"))

traffic = collections.defaultdict(dict)
for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = one(f"prof_{kind}/*/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and ("m355" in r["Kernel_Name"] or "k_final" in r["Kernel_Name"]):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        traffic[k][counter] = sum(v) / len(v)
        traffic[k]["launches_" + counter] = len(v)
out = {}
for k, v in traffic.items():
    fetch_kib, write_kib = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    out[k] = {
        "FETCH_SIZE_KiB_raw": fetch_kib,
        "WRITE_SIZE_KiB_raw": write_kib,
        "read_bytes_corrected": 2.0 * fetch_kib * 1024.0,
        "write_bytes": write_kib * 1024.0,
        "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
        "launches_sampled": v.get("launches_FETCH_SIZE"),
    }
with open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
for k, v in sorted(out.items()):
    print(f"{v['hbm_bytes_per_launch']/1e6:10.1f} MB/launch  {k[:90]}")
for name in ("bench.log", "test.log"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
