"""Every kernel of ONE iteration (between two k_pred_setup launches in the middle of the last solve) from a rocprofv3
--kernel-trace CSV: start, end (us from the iteration's start), queue, grid, name.
usage: trace_iter.py trace.csv [which=-3] [marker=k_pred_setup]   (small LPs have no k_pred_setup launch: use k_fused_residuals)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
marker = sys.argv[3] if len(sys.argv) > 3 else "k_pred_setup"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
i0, i1 = starts[which], starts[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
busy = {}
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("lpipm::", "").replace("void ", "")[:44]
    q = r.get("Queue_Id", "?")
    busy[q] = busy.get(q, 0.0) + (e - s)
    wg = int(r["Workgroup_Size_X"]) or 1
    print(f"{s:9.1f} {e:9.1f} {e-s:8.1f}  q{q}  wgs {int(r['Grid_Size_X'])//wg * int(r['Grid_Size_Y'] or 1) * int(r['Grid_Size_Z'] or 1):6d}  {name}")
print("iteration span", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "us; busy per queue:", {k: round(v, 1) for k, v in busy.items()})
