"""Timeline of the side-by-side factorisation section from a rocprofv3 --kernel-trace CSV: for the last occurrence of
the section (between two k_pred_setup launches) prints every A.D.A^T / update / fix-up launch and every diagonal-block
kernel with start and end relative to the section start, and per queue the busy time."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_pred_setup" in r["Kernel_Name"]]
i0, i1 = starts[-3], starts[-2]          # a full iteration in the middle of the last solve
t0 = int(rows[i0]["Start_Timestamp"])
sec = rows[i0:i1]
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
print("columns:", list(rows[0].keys()))
busy = {}
ndiag = 0
for r in sec:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("lpipm::", "")[-40:]
    q = r.get(qkey, "?") if qkey else "?"
    busy[q] = busy.get(q, 0.0) + (e - s)
    big = any(k in name for k in ("streamk", "fixup"))
    if "potrf_diag" in name:
        ndiag += 1
        if ndiag % 4 == 1: print(f"{s:9.1f} .. {e:9.1f}  q{q}  {name} (#{ndiag})")
    elif big or (e - s) > 40:
        print(f"{s:9.1f} .. {e:9.1f}  q{q}  {name}  [{e-s:.1f} us]")
print("iteration span", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "us; busy per queue:", {k: round(v, 1) for k, v in busy.items()})
