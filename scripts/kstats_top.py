"""Top kernels of a rocprofv3 --stats kernel_stats CSV: calls, average, share.  usage: kstats_top.py file.csv [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms")
for r in rows[:n]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us {100 * int(r['TotalDurationNs']) / tot:5.1f}%  {r['Name'][:100]}")
