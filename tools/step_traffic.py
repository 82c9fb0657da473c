"""HBM traffic of ONE pretraining step, per kernel, from a rocprofv3 kernel trace (which kernels a step launches, in
order) and two rocprofv3 --pmc passes over the same command (FETCH_SIZE, WRITE_SIZE; separate passes).

usage: python tools/step_traffic.py KERNEL_TRACE_CSV FETCH_DIR WRITE_DIR OUT_CSV OUT_JSON

FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM
section), so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-B streaming stores.  The K1 launches of a
step are told apart by their order: student layer 1 / 2 on the augmented graph, teacher layer 1 / 2 on the batch graph.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:70]


def per_dispatch(d, counter):
    """kernel short name -> list of counter values in dispatch order"""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])))
    rows.sort()
    out = collections.defaultdict(list)
    for _, k, v in rows:
        out[k].append(v)
    return out


trace = list(csv.DictReader(open(sys.argv[1])))
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(trace) if "k_ema_lerp" in r["Kernel_Name"]]
step = [short(r["Kernel_Name"]) for r in trace[marks[-3] + 1: marks[-2] + 1]]
counts = collections.Counter(step)
fetch, write = per_dispatch(sys.argv[2], "FETCH_SIZE"), per_dispatch(sys.argv[3], "WRITE_SIZE")


def steady(vals, per_step):
    """average over the dispatches of the last full steps (skips warm-up / set-up dispatches of the same kernel)"""
    if not vals:
        return 0.0
    keep = vals[-per_step * 5:] if len(vals) >= per_step * 5 else vals
    return sum(keep) / len(keep)


rows, total_r, total_w = [], 0.0, 0.0
for k, n in sorted(counts.items(), key=lambda kv: -kv[1]):
    rd = 2 * steady(fetch.get(k, []), n) * 1024
    wr = steady(write.get(k, []), n) * 1024
    rows.append((k, n, rd / 1e6, wr / 1e6, n * (rd + wr) / 1e6))
    total_r += n * rd
    total_w += n * wr
rows.sort(key=lambda r: -r[4])
with open(sys.argv[4], "w") as fh:
    fh.write("# one steady-state pretraining step (C4, bench.py --steps 10 --warmup 3 --no-cpu-baseline --e2e-steps 0):\n")
    fh.write("# launches per step from the kernel trace; per-launch HBM bytes from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE\n")
    fh.write("# (separate passes; read = 2 * FETCH_SIZE KiB on gfx950 per MI355X_MICROARCH.md), averaged over the last 5 steps\n")
    fh.write("kernel,launches_per_step,read_MB_per_launch,write_MB_per_launch,MB_per_step\n")
    for r in rows:
        fh.write('"%s",%d,%.1f,%.1f,%.1f\n' % r)
    fh.write('"TOTAL",%d,%.1f,%.1f,%.1f\n' % (len(step), total_r / 1e6, total_w / 1e6, (total_r + total_w) / 1e6))
k1 = [k for k in counts if k.startswith("stemgnn::k_sage_agg_fwd")]
info = {"traffic_bytes_per_step": total_r + total_w, "read_bytes_per_step": total_r, "write_bytes_per_step": total_w,
        "launches_per_step": len(step),
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) + kernel trace, reduced by "
                  "tools/step_traffic.py; " + os.path.basename(sys.argv[4])}
if k1:
    f, w, n = fetch.get(k1[0], []), write.get(k1[0], []), counts[k1[0]]
    m = min(len(f), len(w)) // n * n
    f, w = f[len(f) - m:], w[len(w) - m:]
    cls = {"augmented": [], "batch": []}
    for s0 in range(0, m, n):  # per step: the launches with the larger fetches ran on the batch graph (the teacher's)
        tot = sorted(((f[i], 2 * f[i] * 1024 + w[i] * 1024) for i in range(s0, s0 + n)), reverse=True)
        cls["batch"] += [t for _, t in tot[:n // 2]]
        cls["augmented"] += [t for _, t in tot[n // 2:]]
    for c, v in cls.items():
        if v:
            info["k1_%s_traffic_bytes_per_launch" % c] = sum(v[-10:]) / len(v[-10:])
json.dump(info, open(sys.argv[5], "w"), indent=1)
print("step traffic: read %.0f MB + write %.0f MB = %.0f MB over %d launches" %
      (total_r / 1e6, total_w / 1e6, (total_r + total_w) / 1e6, len(step)))
