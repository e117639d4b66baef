"""Summarise rocprofv3 --pmc counter_collection csv: per-kernel totals for the biggest dispatch of a named kernel."""
import csv, glob, sys, collections
d, kern = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
by = collections.defaultdict(dict)
for r in rows:
    if kern in r["Kernel_Name"]:
        by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        by[r["Dispatch_Id"]]["_grid"] = int(r["Grid_Size"]); by[r["Dispatch_Id"]]["_vgpr"] = r.get("VGPR_Count"); by[r["Dispatch_Id"]]["_lds"] = r.get("LDS_Block_Size")
best = max(by.values(), key=lambda c: c.get("SQ_WAVE_CYCLES", c.get("SQ_WAVES", 0)) if ("SQ_WAVE_CYCLES" in c or "SQ_WAVES" in c) else sum(v for k, v in c.items() if not k.startswith("_")))
for k in sorted(best):
    print("%-28s %s" % (k, best[k]))
