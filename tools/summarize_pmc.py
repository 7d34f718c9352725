"""Sum a PMC counter per kernel from rocprofv3 --pmc csv output (counter_collection.csv)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for name in ("pmc_fetch", "pmc_write", "pmc_mfma"):
    files = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?").split("(")[0]
            agg[(k, row.get("Counter_Name"))][0] += float(row.get("Counter_Value", 0))
            agg[(k, row.get("Counter_Name"))][1] += 1
    for (k, c), (v, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:40]:
        print(name, c, k[:70], "sum", v, "launches", n, "per launch", v / max(n, 1))
        res["%s|%s" % (c, k)] = {"sum": v, "launches": n}
json.dump(res, open(os.path.join(out, "pmc_raw.json"), "w"), indent=1)
