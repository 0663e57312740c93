"""Derives VALU / MFMA utilisation per kernel from the rocprofv3 --pmc passes of
scripts/gpu_pmc_valu.sh (gpurun_out/pmc_*/**/*_counter_collection.csv) and writes
profiles/<tag>_valu_mfma_util.json.  Formulas (gfx94x derived-counter definitions, which ROCm 7.2
falls back to on gfx950; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles, MI355X_MICROARCH.md):
  valu_busy   = 4 * SQ_ACTIVE_INST_VALU / (n_simd * gui_cycles)      fraction of SIMD-time issuing VALU
  lane_util   = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)   active lanes per VALU instruction (divergence)
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (n_simd * gui_cycles)
with gui_cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs) and n_simd = 1024.
usage: python tools/summarize_pmc.py r01 [pass directory name: only that pass]"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 1024
out = {}
only = sys.argv[2] if len(sys.argv) > 2 else "pmc_*"
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", only))):
    if not os.path.isdir(d):
        continue
    files = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)  # newest pass
    if not files:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        if "m355" in r["Kernel_Name"]:
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in per.items():
        m = {name: sum(v) / len(v) for name, v in c.items()}
        gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        e = {"pass": os.path.basename(d), "launches": len(next(iter(c.values()))), "counters_avg_per_launch": m}
        if gui > 0:
            if "SQ_ACTIVE_INST_VALU" in m:
                e["valu_busy"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (N_SIMD * gui)
            if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_ACTIVE_INST_VALU"):
                e["lane_util"] = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                e["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * gui)
        out[f"{os.path.basename(d)}::{k}"] = e
path = os.path.join(root, "profiles", f"{tag}_valu_mfma_util.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
for k, e in sorted(out.items()):
    print(f"valu_busy {e.get('valu_busy', float('nan')):6.3f} lane_util {e.get('lane_util', float('nan')):6.3f} "
          f"mfma_busy {e.get('mfma_busy', float('nan')):6.3f}  {k[:110]}")
