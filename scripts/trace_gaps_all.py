"""Idle gaps on the main queue for EVERY iteration of a rocprofv3 --kernel-trace CSV (iterations are split at k_pred_setup,
or at another marker): per iteration the span, the busy time and every gap above a threshold with the kernels around it.
usage: trace_gaps_all.py trace.csv [min_gap_us=10] [marker=k_pred_setup]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
marker = sys.argv[3] if len(sys.argv) > 3 else "k_pred_setup"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
name = lambda r: r["Kernel_Name"].split("(")[0].replace("lpipm::", "").replace("void ", "")[:36]
for a, b in zip(starts, starts[1:]):
    it = rows[a:b]
    q = it[0].get("Queue_Id", "?")
    main = [r for r in it if r.get("Queue_Id", "?") == q]
    t0 = int(main[0]["Start_Timestamp"])
    span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    if span > 50000: continue            # between two solves
    gaps = []
    for x, y in zip(main, main[1:]):
        g = (int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3
        if g >= thr: gaps.append(f"{g:.0f} us after {name(x)} @{(int(x['End_Timestamp']) - t0) / 1e3:.0f}")
    print(f"iteration {a:6d}: span {span:8.1f} us, {len(main)} launches; gaps >= {thr:.0f} us: " + ("; ".join(gaps) if gaps else "none"))
