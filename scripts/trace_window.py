"""Kernels of every queue inside a time window of a rocprofv3 --kernel-trace CSV, one line per launch (start, end, duration,
queue, workgroups, name), the window counted from the first launch of the LAST solve's middle.
usage: trace_window.py trace.csv [start_us_from_the_middle=0] [length_us=3500]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
off = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
length = float(sys.argv[3]) if len(sys.argv) > 3 else 3500.0
T0, T1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
t0 = T0 + (T1 - T0) * 3 // 4 + int(off * 1e3)
qs = {}
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if e < 0 or s > length:
        continue
    q = r.get("Queue_Id", "?")
    qs.setdefault(q, len(qs))
    name = r["Kernel_Name"].split("(")[0].replace("lpipm::", "").replace("void ", "")[:40]
    wg = int(r["Workgroup_Size_X"]) or 1
    nwg = int(r["Grid_Size_X"]) // wg * int(r["Grid_Size_Y"] or 1) * int(r["Grid_Size_Z"] or 1)
    print(f"{s:9.1f} {e:9.1f} {e-s:8.1f}  " + "                    " * qs[q] + f"q{q} {nwg:6d} {name}")
