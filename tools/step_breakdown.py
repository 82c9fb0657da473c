"""Summarise one pretraining step from a rocprofv3 kernel trace CSV (kernel_trace.csv)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_ema_lerp" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
seg = rows[a + 1:b + 1]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
print(f"kernels in one step: {len(seg)}  span {span:.3f} ms  busy {busy:.3f} ms")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:78]


c = collections.OrderedDict()
for r in seg:
    k = short(r["Kernel_Name"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    c.setdefault(k, [0, 0])
    c[k][0] += 1
    c[k][1] += d
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for k, (n, d) in sorted(c.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n:4d} {d / 1e3:9.1f} us  {k}")
