"""Print the kernels of one steady-state step in launch order (from a rocprofv3 kernel_trace.csv)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_ema_lerp" in r["Kernel_Name"]]
seg = rows[idx[-3] + 1: idx[-2] + 1]
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::|^void |stemgnn::|at::native::", "", r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{d:8.1f} us  grid {r.get('Grid_Size', '?'):>9}  {n[:110]}")
