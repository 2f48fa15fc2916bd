"""Aggregates rocprofv3 --pmc counter_collection.csv files (one pass per counter group) into the JSON summary kept
under profiles/: per-launch averages for one kernel, HBM-side traffic with the gfx950 FETCH_SIZE correction of
/opt/skills/guides/MI355X_MICROARCH.md, MFMA busy fraction.
usage: pmc_summary.py <kernel substring> <m> <n> <out.json> <dir with *_counter_collection.csv> [...]"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kern, m, n, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
vals = {}
for d in sys.argv[5:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = {}
        for r in csv.DictReader(open(f)):
            if kern not in r["Kernel_Name"]:
                continue
            per_dispatch.setdefault((r["Counter_Name"], r["Dispatch_Id"]), 0.0)
            per_dispatch[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (name, _), v in per_dispatch.items():
            vals.setdefault(name, []).append(v)
# launches that did no work (the head of an iteration enqueued ahead of the status read returns at once when the LP
# turned out to be finished) are not the kernel being measured: drop dispatches below 5 % of the median
import statistics
for k in list(vals):
    med = statistics.median(vals[k])
    vals[k] = [x for x in vals[k] if x >= 0.05 * med]
avg = {k: sum(v) / len(v) for k, v in vals.items()}
from bench import csrc_hash
res = {"kernel": f"{kern} (m={m} n={n})",
       "csrc_sha256": csrc_hash(),     # bench.py reports these counters only while lp_amd/csrc still hashes to this
       "command": "rocprofv3 --pmc <COUNTERS> --output-format csv -- python3 scripts/prof_c3.py   (one pass per counter group)",
       "launches_averaged": {k: len(v) for k, v in vals.items()}}
res.update({k + ("_KiB_raw" if k in ("FETCH_SIZE", "WRITE_SIZE") else ""): v for k, v in avg.items()})
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    res["fetch_correction"] = ("x2: on gfx950 FETCH_SIZE reports 1/2 of the bytes of 16-B-per-lane coalesced loads "
                               "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact")
    res["traffic_bytes_per_launch"] = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0
    nlp = int(os.environ.get("PMC_LPS_PER_LAUNCH", "32" if "units" in kern else "1"))     # a lockstep launch covers the whole batch
    res["algorithmic_bytes_per_launch"] = nlp * (8 * m * n + (4 * m * m if ("streamk" in kern or "units" in kern) else 0))
    res["lps_per_launch"] = nlp
    res["note"] = ("FETCH_SIZE counts L2->fabric requests; Infinity-Cache (256 MiB) hits are included, so this is an upper "
                   "bound on HBM bytes.")
if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg:
    res["mfma_busy_fraction"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    res["mfma_busy_fraction_def"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 [cycles the kernel ran, per XCD] * 1024 SIMDs); "
                                     "64 busy cycles per v_mfma_f64_16x16x4_f64")
if "SQ_INSTS_VALU_MFMA_MOPS_F64" in avg:
    res["mfma_flops_counted"] = avg["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
