"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last solve in it, each kernel's duration and the idle gap
before it (GPU timeline of one IPM iteration): shows whether a small problem is kernel-bound or launch-bound."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last N kernels
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
tail = rows[-N:]
prev_end = None
busy = 0; gap_total = 0
agg = collections.OrderedDict()
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end is not None else 0
    name = r["Kernel_Name"].split("(")[0][-48:]
    print(f"{name:50s} dur {(e-s)/1e3:8.2f} us  gap {gap/1e3:8.2f} us")
    busy += e - s; gap_total += max(gap, 0)
    a = agg.setdefault(name, [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] += max(gap, 0)
    prev_end = e
print(f"busy {busy/1e3:.1f} us, gaps {gap_total/1e3:.1f} us over {len(tail)} kernels")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print(f"  {k:50s} n={a[0]:4d} dur {a[1]/1e3:8.1f} gapbefore {a[2]/1e3:8.1f}")
