"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the same command.
usage: python tools/pmc_traffic.py FETCH_DIR WRITE_DIR OUT_CSV [K1_JSON]
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM
section), so read bytes = 2 * FETCH_SIZE * 1024."""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d, name):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch[k][0] + write[k][0])):
    n = max(fetch[k][1], write[k][1], 1)
    fk = fetch[k][0] / max(fetch[k][1], 1)
    wk = write[k][0] / max(write[k][1], 1)
    rows.append((k[:90], n, fk, 2 * fk * 1024 / 1e6, wk, wk * 1024 / 1e6))
with open(sys.argv[3], "w") as fh:
    fh.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on: python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline\n")
    fh.write("# per-kernel averages over all dispatches; KiB counters; read bytes = 2 * FETCH_SIZE * 1024 on gfx950 (MI355X_MICROARCH.md, HBM)\n")
    fh.write("kernel,dispatches,FETCH_SIZE_KiB_avg,read_MB_corrected,WRITE_SIZE_KiB_avg,write_MB\n")
    for r in rows:
        fh.write('"%s",%d,%.0f,%.1f,%.0f,%.1f\n' % r)
if len(sys.argv) > 4:
    k1 = [r for r in rows if "k_sage_agg_fwd" in r[0]]
    tot_n = sum(r[1] for r in k1)
    rd = sum(r[3] * r[1] for r in k1) / tot_n * 1e6
    wr = sum(r[5] * r[1] for r in k1) / tot_n * 1e6
    json.dump({"kernel": "k_sage_agg_fwd", "workload": "c4 bs1024 fanout [10,10] D=128", "read_bytes_per_launch": rd,
               "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
               "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md; " + os.path.basename(sys.argv[3])}, open(sys.argv[4], "w"), indent=1)
    print("K1 traffic per launch: read %.1f MB write %.1f MB" % (rd / 1e6, wr / 1e6))
