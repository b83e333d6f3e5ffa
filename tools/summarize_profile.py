#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile.sh (gpurun_out/<tag>_*) into profiles/<tag>_kernel_stats.csv
and profiles/<tag>_pmc.csv (one value per counter for the dominant kernel, summed over its dispatches' rows)."""
import csv, glob, os, sys

tag, kernel_substr = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)
for f in glob.glob(os.path.join(out, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(prof, f"{tag}_kernel_stats.csv"), "w", newline="") as g:
        w = csv.writer(g, quoting=csv.QUOTE_ALL)
        for r in rows[:8]:
            w.writerow(r)
vals = {}
for p in ("fetch", "write", "sq", "sq2"):
    for f in glob.glob(os.path.join(out, f"{tag}_pmc_{p}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open(os.path.join(prof, f"{tag}_pmc.csv"), "w") as g:
    g.write(f"# rocprofv3 --pmc passes (separate runs, tools/profile.sh) of: python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ess {' '.join(sys.argv[3:])}\n")
    g.write(f"# kernel: *{kernel_substr}*, ONE launch\n")
    g.write("# FETCH_SIZE/WRITE_SIZE are in KiB as rocprofv3 reports them; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE x2\n")
    g.write("counter,value\n")
    for k in sorted(vals):
        g.write(f"{k},{vals[k]:.0f}\n")
print(vals)
