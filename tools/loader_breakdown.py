"""Device time of the LOADER's kernels per batch (sampler + feature gather + transposed CSR), from a rocprofv3 kernel
trace of `bench.py --e2e-steps N`: every kernel that never occurs inside a steady-state step of the timed loop.
usage: python tools/loader_breakdown.py KERNEL_TRACE_CSV E2E_STEPS"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
short = lambda s: re.sub(r"\(.*$", "", re.sub(r"\(anonymous namespace\)::|^void |stemgnn::", "", s))[:90]
marks = [i for i, r in enumerate(rows) if "k_ema_lerp" in r["Kernel_Name"]]
seg = rows[marks[-n - 1] + 1: marks[-1] + 1]  # the last n steps = the loader-in-loop phase
c = collections.OrderedDict()
for r in seg:
    k = short(r["Kernel_Name"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    c.setdefault(k, [0, 0]); c[k][0] += 1; c[k][1] += d
tot = sum(v[1] for v in c.values())
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
print(f"{n} in-loop steps: span {span / n / 1e3:.1f} us per step, kernel time {tot / n / 1e3:.1f} us per step")
loader = {k: v for k, v in c.items() if re.search(r"sample_hop|k_claim|k_flag_new|k_assign_new|k_emit_edges|k_emit_and_sample|k_count_wins|k_scan_block|k_sort_segments|k_finish|k_seed|rocprim|k_edge_keys|k_rowptr|k_other_endpoint|k_gather|index|k_inv_deg|k_fill_i32|Memset|fillBuffer|copy|elementwise", k)}
lt = sum(v[1] for v in loader.values())
print(f"loader kernels: {lt / n / 1e3:.1f} us per batch in {sum(v[0] for v in loader.values()) / n:.1f} launches")
for k, (cnt, d) in sorted(loader.items(), key=lambda kv: -kv[1][1]):
    print(f"{cnt / n:6.1f} x {d / cnt / 1e3:7.1f} us = {d / n / 1e3:7.1f} us/batch  {k}")

# the step's own kernels: resident phase (the n steps before) against the in-loop phase, by kernel
res = rows[marks[-2 * n - 1] + 1: marks[-n - 1] + 1]
def by_kernel(seg_rows):
    d = collections.OrderedDict()
    for r in seg_rows:
        k = short(r["Kernel_Name"])
        if k in loader:
            continue
        d.setdefault(k, [0, 0]); d[k][0] += 1; d[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d
a, b = by_kernel(res), by_kernel(seg)
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
span_a = int(res[-1]["End_Timestamp"]) - int(res[0]["Start_Timestamp"])
print(f"\nstep kernels: resident {ta / n / 1e3:.1f} us/step (span {span_a / n / 1e3:.1f}), in-loop {tb / n / 1e3:.1f} us/step (span {span / n / 1e3:.1f})")
diff = sorted(((b.get(k, [0, 0])[1] - v[1]) / n / 1e3, k, v[0] / n, v[1] / max(v[0], 1) / 1e3, b.get(k, [0, 1])[1] / max(b.get(k, [1, 0])[0], 1) / 1e3)
              for k, v in a.items())
for dlt, k, cnt, ua, ub in diff[::-1][:14]:
    print(f"  {dlt:+7.1f} us/step  {cnt:4.1f} x {ua:7.1f} -> {ub:7.1f} us  {k}")
qs = collections.Counter(r.get("Queue_Id", "?") for r in seg)
print("launches per queue in the in-loop phase:", dict(qs))

# gaps on the step's queue in the in-loop phase, and what the loader's queue was doing during them
main_q = qs.most_common(1)[0][0]
mq = [r for r in seg if r.get("Queue_Id") == main_q]
sq = [r for r in seg if r.get("Queue_Id") != main_q]
def gaps_of(rs):
    g = []
    for x, y in zip(rs, rs[1:]):
        d = int(y["Start_Timestamp"]) - int(x["End_Timestamp"])
        g.append((d, x, y))
    return g
for label, rs in (("resident", [r for r in res]), ("in-loop", mq)):
    g = gaps_of(rs)
    tot = sum(d for d, _, _ in g if d > 0)
    big = [d for d, _, _ in g if d > 5000]
    print(f"{label}: idle between the step's kernels {tot / n / 1e3:.1f} us/step; gaps > 5 us: {len(big) / n:.1f} per step, {sum(big) / n / 1e3:.1f} us/step")
g = sorted(gaps_of(mq), key=lambda t: -t[0])[:12]
for d, x, y in g:
    a0, a1 = int(x["End_Timestamp"]), int(y["Start_Timestamp"])
    during = [short(r["Kernel_Name"])[:28] for r in sq if int(r["Start_Timestamp"]) < a1 and int(r["End_Timestamp"]) > a0]
    print(f"  gap {d / 1e3:7.1f} us after {short(x['Kernel_Name'])[:34]} before {short(y['Kernel_Name'])[:34]}; loader queue: {during[:4]}")
