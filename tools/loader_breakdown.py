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
loader = {k: v for k, v in c.items() if re.search(r"sample_hop|k_claim|k_flag_new|k_assign_new|k_emit_edges|k_finish|k_seed|rocprim|k_edge_keys|k_rowptr|k_other_endpoint|k_gather|index|k_inv_deg|k_fill_i32|Memset|fillBuffer|copy|elementwise", k)}
lt = sum(v[1] for v in loader.values())
print(f"loader kernels: {lt / n / 1e3:.1f} us per batch in {sum(v[0] for v in loader.values()) / n:.1f} launches")
for k, (cnt, d) in sorted(loader.items(), key=lambda kv: -kv[1][1]):
    print(f"{cnt / n:6.1f} x {d / cnt / 1e3:7.1f} us = {d / n / 1e3:7.1f} us/batch  {k}")
