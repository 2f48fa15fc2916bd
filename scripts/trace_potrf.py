"""Timeline of one factorisation from a rocprofv3 --kernel-trace CSV (last complete iteration): diag kernels and the
K = 512 trailing updates per queue."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_pred_setup" in r["Kernel_Name"]]
i0, i1 = starts[-3], starts[-2]
t0 = int(rows[i0]["Start_Timestamp"])
nd = 0
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("lpipm::", "").replace("void ", "")[-36:]
    if "potrf_diag" in name:
        nd += 1
        if nd % 4 in (1, 0): print(f"{s:9.1f} .. {e:9.1f} q{r['Queue_Id']} diag #{nd}")
    elif "gemm_nt_tile_kernel" in name or "grouped" in name or "streamk" in name:
        print(f"{s:9.1f} .. {e:9.1f} q{r['Queue_Id']} {name} [{e-s:.1f}] grid {r['Grid_Size_X']}")
