"""Timeline of ONE factorisation (lpipm_k_potrf hook) from a rocprofv3 --kernel-trace CSV: every kernel of the last call,
start .. end in us relative to its first kernel, with its queue.  usage: trace_factor.py kernel_trace.csv [ncalls=3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# calls are separated by the D2D copy of the matrix; find the diag kernels with global_row0 == 0 ... simpler: split at gaps > 200 us
starts = [0]
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 150000: starts.append(i)
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("lpipm::", "").replace("void ", "")
    print(f"{s:9.1f} .. {e:9.1f} [{e - s:7.1f}] q{r['Queue_Id']} {name[:44]} grid {r['Grid_Size_X']}")
