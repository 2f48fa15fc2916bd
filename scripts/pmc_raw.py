"""Per-launch averages of every counter in rocprofv3 --pmc CSVs for one kernel: pmc_raw.py <kernel substring> <dir> [...]"""
import csv, glob, os, sys, statistics
kern = sys.argv[1]
vals = {}
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                per[(r["Counter_Name"], r["Dispatch_Id"])] = per.get((r["Counter_Name"], r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        for (name, _), v in per.items():
            vals.setdefault(name, []).append(v)
for k in sorted(vals):
    med = statistics.median(vals[k]); v = [x for x in vals[k] if x >= 0.05 * med] or vals[k]
    print(f"{k:32s} {sum(v)/len(v):16.1f}  ({len(v)} launches)")
