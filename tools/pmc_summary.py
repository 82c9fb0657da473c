"""Average rocprofv3 --pmc counters per kernel name from counter_collection.csv files.
usage: python tools/pmc_summary.py DIR"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:110] + "|grid" + r.get("Grid_Size", "")
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print(f"   {c:32s} {s / n:16.0f}  (n={n})")
