#!/usr/bin/env python3
"""Headline benchmark: pretrain edges/sec (fwd+bwd) on the 1M-node / 20M-edge synthetic graph
(BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full pretraining iteration of reference pretrain.py:41-66 (augment -> student
forward -> VQ -> 4 reconstruction losses incl. the teacher forward -> backward -> clip ->
AdamW -> cosine LR -> EMA teacher) on one neighbour-sampled mini-batch (fan-out [10,10], 1024
seeds per rank) whose inputs are already resident in HBM.  value = sum over ranks and steps of
the un-augmented batch graph's edge count / max-over-ranks wall time.  fp32 throughout.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (num_nodes, num_edges, dim, edge types, full_batch)
    "c4": dict(nodes=1_000_000, edges=20_000_000, dim=128, types=4, full_batch=False,
               desc="C4: 1M-node/20M-edge synthetic Graph-U, neighbour-sampled [10,10], 1024 seeds/rank"),
    "c2": dict(nodes=100_000, edges=1_000_000, dim=128, types=4, full_batch=True,
               desc="C2: 100k-node/1M-edge synthetic Graph-U, full batch"),
    # stand-ins for the configs whose data cannot be materialised offline: same node / edge /
    # feature / codebook sizes, synthetic structure and unit-norm features; fp32 like the reference (no autocast)
    "c3": dict(nodes=169_343, edges=2_315_598, dim=768, types=1, full_batch=True, codebook=512,
               desc="C3 stand-in: ogbn-arxiv-sized (169,343 nodes, 2,315,598 directed entries after ToUndirected), "
                    "D=768, K=512, full batch"),
    # the reference's OWN default width on the sampled C4 graph (config/pretrain.yaml:3-20: hidden_dim = code_dim = 768,
    # codebook_size 128, 4 heads, batch 1024; pretrain.py:151-153 NeighborLoader [10, 10])
    "refdefault": dict(nodes=1_000_000, edges=20_000_000, dim=768, types=4, full_batch=False, codebook=128,
                       desc="reference default width: 1M-node/20M-edge synthetic Graph-U, D = code_dim = 768, H = 4, K = 128, "
                            "neighbour-sampled [10,10], 1024 seeds/rank"),
    "c5": dict(nodes=0, edges=0, dim=768, types=0, full_batch=False, codebook=2048, mix="all",
               desc="C5 stand-in: --pretrain_dataset all as a union of nine synthetic member graphs with the real "
                    "datasets' node / edge / text-row / edge-type counts (molecule sets scaled to 1.2M nodes), seeds "
                    "weighted per member (config/pt_data.yaml) and rebuilt per epoch, D=768, K=2048, "
                    "neighbour-sampled [10,10], 1024 seeds/rank"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--codebook-size", type=int, default=0, help="0 = the workload's (128 unless it says otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--e2e-steps", type=int, default=40, help="extra steps timed with the loader inside the loop (0: skip)")
    ap.add_argument("--feature-dtype", default="auto", choices=["auto", "f32", "bf16"],
                    help="storage of node features / layer outputs (auto: bf16 for c5 -- BASELINE config 5 -- else f32)")
    ap.add_argument("--gemm", default="auto", choices=["auto", "f32", "bf16"],
                    help="dense products: f32 = fp32 results from exact bf16 pieces (the headline mode); bf16 = one bf16 "
                         "matrix pass on rounded operands, fp32 accumulation, fp32 VQ core (BASELINE config 5); auto = "
                         "bf16 for the c5 workload, f32 otherwise")
    ap.add_argument("--pmc-traffic", default="auto", choices=["auto", "off"],
                    help="auto (N = 1, c4): measure roofline.traffic live -- two short child runs of this script under "
                         "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); off: quote profiles/step_traffic.json")
    ap.add_argument("--preheat", default="auto", choices=["auto", "off"],
                    help="auto: untimed steps until the step time has settled (>= 0.3 s, <= 1 s), before the counted "
                         "--warmup; off: the counted warm-up only")
    ap.add_argument("--no-dense-profile", action="store_true",
                    help="do not stamp the big-tile core's launches with HIP events in the timed region (A/B of the stamps' cost)")
    ap.add_argument("--no-extra", action="store_true", help="skip the K1 legs at C2 / C3 size and the dense-product leg")
    ap.add_argument("--grad-sync", default="flat", choices=["flat", "ddp"],
                    help="--gpus > 1: flat = one fused copy + one all-reduce per step (parallel.FlatGradSync); ddp = "
                         "DistributedDataParallel's reducer (three buckets overlapped with the backward)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    return ap.parse_args()


def cpu_baseline(params, batch_cpu, bs, budget_s):
    """The CPU oracle (restated reference path) timed on this box's host cores: same step,
    same batch shape, fp32.  kind = "port" (the reference's PyG path cannot run here)."""
    from oracle import stem_oracle as O
    x, ei, table, et = batch_cpu
    cores = int(os.environ.get("STEMGNN_CPU_THREADS", min(os.cpu_count() or 1, 16)))  # the box's CPU share per GPU
    torch.set_num_threads(cores)
    D = x.size(1)
    torch.manual_seed(42)
    om = O.build_oracle_model(D, params["num_layers"], params["codebook_head"], params["codebook_size"],
                              params["code_dim"], dropout=params["dropout"])
    opt = torch.optim.AdamW(om.parameters(), lr=params["pretrain_lr"], weight_decay=params["pretrain_weight_decay"])
    n, e = x.size(0), ei.size(1)
    ea = table[et]
    g = torch.Generator().manual_seed(0)

    def draws():
        keep = torch.rand(e, generator=g) >= params["edge_p"]
        es = max(int(e * params["topo_recon_ratio"]), 1)
        return {
            "feat_keep": torch.rand(D, generator=g) >= params["feat_p"],
            "edge_keep": keep,
            "student_dropout": [torch.rand(n, D, generator=g) >= params["dropout"] for _ in range(params["num_layers"] - 1)],
            "teacher_dropout": [torch.rand(n, D, generator=g) >= params["dropout"] for _ in range(params["num_layers"] - 1)],
            "topo_perm": torch.randperm(e, generator=g)[:es],
            "neg_edge_index": torch.randint(0, n, (2, es), generator=g),
            "topo_sem_perm": torch.randperm(e, generator=g)[:es],
            "ortho_ids": torch.randperm(params["codebook_size"], generator=g)[:params["ortho_reg_max_codes"]],
        }

    O.pretrain_step(om, opt, None, params, x, ei, ea, bs, draws())  # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    steps = 0
    while True:
        O.pretrain_step(om, opt, None, params, x, ei, ea, bs, draws())
        steps += 1
        if time.perf_counter() - t0 >= budget_s or steps >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": e * steps / dt, "unit": "edges/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{steps} full pretrain steps of the CPU oracle on one batch ({n} nodes, {e} edges, D={D}), "
                      f"{dt:.1f} s, torch {torch.__version__} fp32, {cores} threads on {cpu_model()}"}


def cpu_model() -> str:
    """The host CPU as /proc/cpuinfo names it (SURVEY.md section 8d asks for the core count AND the model)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def pmc_step_traffic(args, timeout_s=300):
    """HBM bytes from the PMC counters, collected as MI355X_MICROARCH.md prescribes: one rocprofv3 --pmc pass per counter
    (FETCH_SIZE, WRITE_SIZE), nothing else traced, over a short run of THIS script (child processes; this process never
    execs).  FETCH_SIZE tallies 128-B requests at 64 B on gfx950: read bytes = 2 * FETCH_SIZE KiB; WRITE_SIZE is exact for
    16-byte streaming stores.  Steps are cut at the teacher's EMA kernel; the K1 launches of a step are split into the
    two on the batch graph (the larger fetches) and the two on the augmented graph.
    -> dict(traffic_bytes_per_step, k1_batch_traffic_bytes_per_launch, k1_augmented_traffic_bytes_per_launch) or
    dict(error=...)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    # under somebody else's profiler (its tool library preloaded into this process and inherited by children) a nested
    # rocprofv3 would initialise the GPU in its launcher and then exec: not from here
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {"error": "this run is itself being profiled"}
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="stemgnn_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", counter, "--", sys.executable,
               os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extra",
               "--e2e-steps", "0", "--pmc-traffic", "off", "--preheat", "off", "--workload", args.workload, "--batch-size",
               str(args.batch_size), "--gemm", args.gemm, "--feature-dtype", args.feature_dtype]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=timeout_s)
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == counter:
                            rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
            rows.sort()
            if r.returncode != 0 or not rows:
                return {"error": f"rocprofv3 --pmc {counter}: rc {r.returncode}, {len(rows)} counter rows"}
            per[counter] = rows
        except (subprocess.TimeoutExpired, OSError) as e:
            return {"error": f"rocprofv3 --pmc {counter}: {type(e).__name__}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)

    def steps_of(rows, scale):
        """bytes per dispatch, cut into steps at k_ema_lerp (the step's last kernel); the last full steps only"""
        cuts = [i for i, (_, k, _) in enumerate(rows) if "k_ema_lerp" in k]
        return [[(k, v * 1024.0 * scale) for _, k, v in rows[a + 1:b + 1]] for a, b in zip(cuts[:-1], cuts[1:])][-4:]

    rd, wr = steps_of(per["FETCH_SIZE"], 2.0), steps_of(per["WRITE_SIZE"], 1.0)
    if not rd or len(rd) != len(wr) or any(len(a) != len(b) for a, b in zip(rd, wr)):
        return {"error": "the two counter passes saw different launch sequences"}
    totals, k1b, k1a = [], [], []
    for a, b in zip(rd, wr):
        totals.append(sum(v for _, v in a) + sum(v for _, v in b))
        k1 = sorted(((va, va + vb) for (ka, va), (_, vb) in zip(a, b) if "k_sage_agg_fwd" in ka), reverse=True)
        half = len(k1) // 2
        k1b += [t for _, t in k1[:half]]
        k1a += [t for _, t in k1[half:]]
    mean = lambda v: sum(v) / len(v) if v else None  # noqa: E731
    return {"traffic_bytes_per_step": mean(totals), "k1_batch_traffic_bytes_per_launch": mean(k1b),
            "k1_augmented_traffic_bytes_per_launch": mean(k1a), "launches_per_step": len(rd[-1]),
            "source": "measured in this run: two child runs of bench.py (6 steps) under rocprofv3 --pmc FETCH_SIZE and "
                      "--pmc WRITE_SIZE (separate passes, nothing else traced); read = 2 x FETCH_SIZE KiB on gfx950 "
                      "(MI355X_MICROARCH.md), averaged over the last four steps"}


def step_algorithmic_bytes(N, A, E, E_aug, k, bs, D, HD, in_dim, T):
    """Algorithmic HBM bytes of one pretraining step as this library runs it (DESIGN.md section 5 lists the same table):
    what every kernel must read and write once, fp32 activations, int32 structure.  SURVEY.md section 8d's form is
    2 B_K1(E) + 2 B_K1(E_aug) + B_K2 + decoders + c N D 4; the entries below are that sum with c spelled out."""
    ND, AD, NH = N * D * 4, A * D * 4, N * HD * 4

    def b_k1(e):  # type-indexed edge attribute, rows = the A rows that can receive edges
        return e * D * 4 + (4 * e + T * D * 4) + 4 * e + 4 * (A + 1) + AD

    t = {
        "augment: mask_feature (read x, write x_aug) + dropout_adj on the CSR": 2 * ND + 32 * E,
        "student K1 x2 (augmented graph)": 2 * b_k1(E_aug),
        "teacher K1 x2 (batch graph)": 2 * b_k1(E),
        "student layer products x2 (read h + agg, write y)": 2 * (2 * ND + AD),
        "student BatchNorm/act/dropout x2 (read y, write h)": 2 * 2 * ND,
        "teacher layer 1 product + normalise": (2 * ND + AD) + 2 * ND,
        "teacher layer 2 product (statistics over all rows, seed rows written)": ND + AD + 3 * bs * D * 4,
        "VQ project_in (read z, write xp)": ND + NH,
        "VQ assignment (read xp; ind, norm)": NH + 12 * N * (HD // D),
        "VQ project_out off the code table (write quantize)": ND + 8 * N * (HD // D),
        "heads forward: lin(q) for the topology head, gathers of 3k edge endpoints": 2 * ND + 8 * k * D * 4,
        "heads backward: zero + scatter + lin backward-data + weight gradient": ND + 2 * ND + 2 * ND + 10 * k * D * 4,
        "seed-row heads (feat, sem) forward + backward": 12 * bs * max(D, in_dim) * 4,
        "VQ backward: code segment sums (read g_quantize)": ND + 8 * N * (HD // D),
        "VQ backward: assignment + project_out backward-data, fused (read g, xp; write g_xp)": ND + 2 * NH,
        "VQ backward: project_in backward-data (read g_xp, write g_z)": NH + ND,
        "VQ backward: project_in weight gradient (read g_xp, z)": NH + ND,
        "encoder backward layer 2: BatchNorm sums, apply, 2 weight gradients, 2 backward-data, K2": 2 * ND + 3 * ND + 2 * ND + 2 * AD + 2 * ND + 2 * AD + (E_aug * D * 4 + 3 * ND),
        "encoder backward layer 1: BatchNorm sums, apply, 2 weight gradients": 2 * ND + 3 * ND + 2 * ND + 2 * AD,
    }
    return t


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` as a plain command: start N rank processes through torch.distributed.run (the
    launch line the driver itself uses) from THIS process, which has made no HIP call (device_count() does not
    initialise the runtime on this image), and pass their exit code on.  With fewer than N devices one JSON line
    says so and the exit code is 0 (nothing to measure is not a failure of the path)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("STEMGNN_SHARE_DEVICE"):
        print(json.dumps({"metric": "pretrain edges/sec (fwd+bwd) on 1M-node/20M-edge synthetic graph, 1/2/4/8 GPUs",
                          "value": None, "unit": "edges/s", "n_gpus": args.gpus, "skipped": True,
                          "reason": f"--gpus {args.gpus} requested, {have} device(s) visible"}), flush=True)
        return 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def k1_at_full_graph_sizes(dev):
    """K1 forward at the full-batch sizes of BASELINE configs 2 and 3 (working sets beyond the Infinity Cache), timed
    with the library's per-launch HIP events; reported beside the C4 batch launches of `roofline`."""
    from stem_gnn_amd import ops
    from stem_gnn_amd.data.synthetic import make_graph
    from stem_gnn_amd.graph import EdgeTypeAttr, GraphStructure
    out = {}
    for name, (n, e, d, t) in {"c2 (100k nodes, 1M edges, D=128)": (100_000, 1_000_000, 128, 4),
                               "c3 stand-in (169,343 nodes, 2,315,598 edges, D=768)": (169_343, 2_315_598, 768, 1)}.items():
        g = make_graph(n, e, d, t, kind="U", device=dev, graph_seed=1234, feat_seed=0)
        x = g.node_text_feat if g.node_text_feat.size(0) == n else g.node_text_feat[g.x]
        gs = GraphStructure(g.edge_index, n, g.xe, validate=True)
        for _ in range(3):
            ops.sage_agg_fwd(x, gs, None, g.edge_text_feat)
        ops.k1_timer.reset(True)
        for _ in range(10):
            ops.sage_agg_fwd(x, gs, None, g.edge_text_feat)
        torch.cuda.synchronize()
        ms, launches, by = ops.k1_timer.collect()
        ops.k1_timer.reset(False)
        gbs = by / (ms * 1e-3) / 1e9
        out[name] = {"kernel": "k_sage_agg_fwd", "edge_attr": "type-indexed (4E + T*D*4 bytes)",
                     "avg_launch_us": ms * 1e3 / launches,
                     "algorithmic_bytes_per_launch": by / launches, "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                     "frac": gbs / 8000.0}
        if n == 100_000:
            # the drop-in signature (reference pretrain.py:38 hands the operator a dense [E, D] edge_attr =
            # edge_text_feat[xe]; encoder.py:72-73): the same kernel reading an [E, D] row per edge through the
            # original edge id -- SURVEY.md section 8(d)'s 1 080 B/edge form
            dense = g.edge_text_feat[g.xe].contiguous()
            for _ in range(3):
                ops.sage_agg_fwd(x, gs, dense, None)
            ops.k1_timer.reset(True)
            for _ in range(10):
                ops.sage_agg_fwd(x, gs, dense, None)
            torch.cuda.synchronize()
            ms, launches, by = ops.k1_timer.collect()
            ops.k1_timer.reset(False)
            gbs = by / (ms * 1e-3) / 1e9
            out[name + ", dense edge_attr [E, D] (drop-in signature)"] = {
                "kernel": "k_sage_agg_fwd", "edge_attr": "dense (E*D*4 + 4E bytes)", "avg_launch_us": ms * 1e3 / launches,
                "algorithmic_bytes_per_launch": by / launches, "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                "frac": gbs / 8000.0}
            del dense
        del g, x, gs
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        return spawn_ranks(args)  # plain `python bench.py --gpus N`: this process never touches the GPU
    if os.environ.get("STEMGNN_SHARE_DEVICE"):  # rehearsal only: several ranks on one card (gloo backend)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)

    from stem_gnn_amd import ops
    from stem_gnn_amd.data.sampler import HipNeighborSampler, NeighborLoader
    from stem_gnn_amd.data.synthetic import make_graph
    from stem_gnn_amd.graph import EdgeTypeAttr, GraphStructure, set_validation
    from stem_gnn_amd.pretrain import build_model, build_optimizer, default_params, pretrain_step
    from stem_gnn_amd.utils.others import seed_everything

    wl = WORKLOADS[args.workload]
    D = wl["dim"]
    params = default_params()
    params.update(input_dim=D, hidden_dim=D, code_dim=D,
                  codebook_size=args.codebook_size or wl.get("codebook", 128), pretrain_batch_size=args.batch_size)
    seed_everything(params["seed"])

    # ---- data: identical synthetic graph on every rank (replicated structure + features)
    if wl.get("mix"):
        from stem_gnn_amd.data.multi import mix_weights, synthetic_mix
        g = synthetic_mix(wl["mix"], dim=D, device=dev, seed=1234)
        wl = dict(wl, nodes=g.num_nodes, edges=int(g.edge_index.size(1)), types=int(g.edge_text_feat.size(0)))
    else:
        g = make_graph(wl["nodes"], wl["edges"], D, wl["types"], kind="U", device=dev, graph_seed=1234, feat_seed=0,
                       feat_rows=wl.get("feat_rows", 0))
    gemm_bf16 = args.gemm == "bf16" or (args.gemm == "auto" and args.workload == "c5")
    if gemm_bf16:
        ops.linear_set_mode(2)  # every Linear of the path as one bf16 matrix pass; the VQ similarity core stays exact
    feat_bf16 = args.feature_dtype == "bf16" or (args.feature_dtype == "auto" and args.workload == "c5")
    if feat_bf16:
        g.node_text_feat = g.node_text_feat.bfloat16()  # the feature table lives in HBM as bf16; arithmetic stays fp32
    total = args.steps + args.warmup
    batches = []
    # The loader hands every batch over in the kernels' native layout (both CSR views + edge types
    # per slot), like the reference's NeighborLoader hands over its own; this is data-pipeline work
    # outside the measured step (inputs resident on the device).
    if wl["full_batch"]:
        x = g.node_text_feat if g.node_text_feat.size(0) == wl["nodes"] else g.node_text_feat[g.x]
        gs = GraphStructure(g.edge_index, wl["nodes"], g.xe, validate=True).ensure_transpose()
        for _ in range(total):
            batches.append((x, gs, g.xe, wl["nodes"]))
    else:
        sampler = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat,
                              [10] * params["num_layers"], seed=100 + rank)
        if wl.get("mix"):
            from stem_gnn_amd.data.sampler import MixLoader
            loader = MixLoader(sampler, g.ptr, list(mix_weights(wl["mix"]).values()), args.batch_size, rank=rank,
                               world_size=world, seed=7, device=dev)
        else:
            loader = NeighborLoader(sampler, torch.arange(g.num_nodes, device=dev), args.batch_size, shuffle=True,
                                    rank=rank, world_size=world, seed=7)
        def epochs():  # batch stream that starts a new epoch when a rank's shard runs out (same count on all ranks)
            while True:
                yield from loader

        it = epochs()
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        for _ in range(total):
            b = next(it)  # fused HIP sampler: batch arrives with its by-target CSR
            x = ops.gather_rows(g.node_text_feat, b.x.contiguous(), validate=False)  # node_text_feat[data.x], on device
            batches.append((x, b.graph.ensure_transpose(), b.xe, b.batch_size))
        torch.cuda.synchronize()
        sampler_ms = (time.perf_counter() - t_s) / total * 1e3
    torch.cuda.synchronize()
    if wl["full_batch"]:
        sampler_ms = 0.0

    # ---- model
    model = build_model(params, dev)
    for p in model.sem_encoder.parameters():
        p.requires_grad_(False)  # never receives gradients (pt_model.py:93 detach); keeps DDP bucketing exact
    opt, sched = build_optimizer(model, params)
    fwd, sync = None, None
    if world > 1:
        # gradient exchange: one fused copy + ONE RCCL all-reduce (AVG) per step behind the backward
        # (parallel.FlatGradSync: +0.1 % on the step at world size 1) or DistributedDataParallel's reducer (three
        # buckets overlapped with the backward, +8 %: configs_extra.dp1_under_ddp measures both on one card)
        if args.grad_sync == "ddp":
            from stem_gnn_amd.parallel import wrap_ddp
            fwd = wrap_ddp(model, local_rank)
        else:
            from stem_gnn_amd.parallel import FlatGradSync
            sync = FlatGradSync(model.parameters())
    set_validation(False)  # batches come from our own sampler: skip the per-build device->host range check
    model.train()

    def step(i):
        x, ei, xe, bs = batches[i]
        return pretrain_step(model, opt, sched, params, x, ei, EdgeTypeAttr(g.edge_text_feat, xe), bs,
                             record_draws=False, forward_fn=fwd, grad_sync=sync)

    # Housekeeping BEFORE the warm-up steps, so that the device goes from the last warm-up step straight into the timed
    # ones (a 100 ms host pause between them lets the clocks drop, and a 20-step timed region is then 3 % slower than
    # a 100-step one):
    #  * a generation-2 cyclic GC pass over the long-lived objects (model, resident batches) costs ~70 ms when it
    #    happens to fire inside a step: collect now, then keep the collector off the hot loop;
    #  * the caching allocator's pool is reserved in one piece: batch sizes differ by a fraction of a per cent, and a
    #    step whose largest buffer is the largest seen so far would otherwise go to hipMalloc inside the timed region.
    gc.collect()
    gc.freeze()
    gc.disable()
    if not wl["full_batch"]:
        pool = torch.empty(6 << 30, dtype=torch.uint8, device=dev)
        del pool
    # Time-based pre-heat (disclosed as config.preheat_steps / preheat_s; untimed, in ADDITION to the counted --warmup):
    # on a fresh lease five warm-up steps are 8 ms -- the clocks of a device that has idled through graph generation
    # have not ramped by then and the first timed steps pay for it (round 3: 1.871 ms on the driver's fresh box
    # against 1.50-1.60 for the same process a second later).  Untimed steps run in windows of five until at least
    # PREHEAT_MIN_S have passed AND two consecutive windows agree within 2 %, capped at PREHEAT_MAX_S.  Under DDP
    # every rank runs the same number of windows (the decision is all-reduced), or the reducers would not pair up.
    PREHEAT_MIN_S, PREHEAT_MAX_S, WIN = 0.3, 1.0, 5
    preheat_steps, t_pre, prev_win = 0, time.perf_counter(), None
    while args.preheat != "off":
        torch.cuda.synchronize()
        tw = time.perf_counter()
        for j in range(WIN):
            step((preheat_steps + j) % total)
        torch.cuda.synchronize()
        now = time.perf_counter()
        win = now - tw
        preheat_steps += WIN
        settled = prev_win is not None and abs(win - prev_win) <= 0.02 * min(win, prev_win)
        prev_win = win
        more = 0.0 if (now - t_pre >= PREHEAT_MAX_S or (now - t_pre >= PREHEAT_MIN_S and settled)) else 1.0
        if world > 1:
            flag = torch.tensor([more], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            more = float(flag.item())
        if more == 0.0:
            break
    preheat_s = time.perf_counter() - t_pre
    for i in range(args.warmup):
        step(i)
    ops.k1_timer.reset(True)
    ops.bigtile_profile(not args.no_dense_profile)  # no-op for the D = 128 workloads: the big-tile core takes the D >= 256 products only
    # one HIP event in front of every timed step and one behind the last, on the stream the step's kernels run on
    # (the library launches on torch's current stream): step_ms / step_ms_median beside the wall mean
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for ev in marks:
        ev.record()  # (an event object creates its hipEvent on first use: not inside the timed region)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        marks[i - args.warmup].record()
        step(i)
    marks[args.steps].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    gc.enable()
    k1_rows = ops.k1_timer.collect_each()
    ops.k1_timer.reset(False)
    ops.bigtile_profile(False)
    bt_ms, bt_flop, bt_launches = ops.bigtile_profile_collect()

    from stem_gnn_amd.parallel import reduce_bench_stats
    edges = float(sum(batches[i][1].num_edges for i in range(args.warmup, total)))
    dt, edges = reduce_bench_stats(dt, edges, dev)

    # the same step with the loader INSIDE the loop (sample with both CSR views -> gather features ->
    # step), reported beside the headline, never as `value`.  The loader runs two batches ahead on a side stream
    # (data/sampler.py PrefetchLoader): the host never waits for a batch's sizes.
    e2e_ms = None
    if not wl["full_batch"] and args.e2e_steps > 0:
        from stem_gnn_amd.data.sampler import PrefetchLoader

        def prepare(b):  # both CSR views, 1 / in-degree and the row ids come from the sampler's own launches
            b.feat = ops.gather_rows(g.node_text_feat, b.x, validate=False, capacity=b.cap_nodes)

        class Rest:  # this rank's loader from the top of its shard again, epoch after epoch
            def iter_pending(self):
                while True:
                    yield from loader.iter_pending()

            def __len__(self):
                return 0

        def in_loop(steps, warm=5):
            # `warm` untimed steps first: the loader's side stream gets its hardware queue and its own allocator pool
            # on first use (a few hipMallocs: ~7 ms, once per run, which would otherwise be spread over `steps`)
            done, t1 = 0, None
            for b in PrefetchLoader(Rest(), dev, prepare):
                if done == warm:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                if done == warm + steps:
                    break
                pretrain_step(model, opt, sched, params, b.feat, b.graph, EdgeTypeAttr(g.edge_text_feat, b.xe), b.batch_size,
                              record_draws=False, forward_fn=fwd, grad_sync=sync)
                done += 1
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / max(done - warm, 1) * 1e3

        e2e_ms = in_loop(args.e2e_steps)


    # Multi-GPU readiness proxy (no multi-GPU lease in this pool): the SAME step with the model under
    # DistributedDataParallel at world size 1 on the nccl (= RCCL) backend -- the reducer's hooks, its bucket views (three
    # buckets per step, parallel.ddp_bucket_cap_mb), FusedAdamW reading them -- so its ms_per_step beside the plain one is
    # DDP's cost on the real step with no peer to wait for.  Runs LAST (it re-homes the gradients into bucket views).
    ddp_proxy = None
    if world == 1 and not wl["full_batch"] and not args.no_extra and args.workload == "c4":
        try:
            import socket
            from stem_gnn_amd.parallel import wrap_ddp
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            import datetime
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev,
                                    timeout=datetime.timedelta(seconds=60))
            ddp_proxy = {"backend": "nccl (RCCL), world size 1", "steps": args.steps,
                         "note": "same step, model under DistributedDataParallel (reducer hooks, gradient bucket views, "
                                 "FusedAdamW on them); no peer: what DDP itself costs on this step"}
            for label, cap in (("one_bucket", 25.0), ("three_buckets", None)):  # torch's default cap, then wrap_ddp's own
                fwd1 = wrap_ddp(model, local_rank, bucket_cap_mb=cap)

                def ddp_step(i):
                    x, ei, xe, bs = batches[i % total]
                    return pretrain_step(model, opt, sched, params, x, ei, EdgeTypeAttr(g.edge_text_feat, xe), bs,
                                         record_draws=False, forward_fn=fwd1)

                for i in range(10):  # the first step runs as one bucket; the reducer re-cuts its buckets after it
                    ddp_step(i)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    ddp_step(args.warmup + i)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / args.steps * 1e3
                log = fwd1._get_ddp_logging_data()
                sizes = str(log.get("rebuilt_bucket_sizes", "") or log.get("bucket_sizes", ""))
                ddp_proxy[label] = {"ms_per_step": ms, "buckets_per_step": len([v for v in sizes.split(",") if v.strip()]),
                                    "bucket_bytes": sizes}
                del fwd1
            ddp_proxy["ms_per_step"] = ddp_proxy["three_buckets"]["ms_per_step"]  # wrap_ddp's default
            # the explicit exchange bench.py uses for --gpus N (parallel.FlatGradSync: one fused copy + one all-reduce)
            from stem_gnn_amd.parallel import FlatGradSync
            sync = FlatGradSync(model.parameters())

            def flat_step(i):
                x, ei, xe, bs = batches[i % total]
                return pretrain_step(model, opt, sched, params, x, ei, EdgeTypeAttr(g.edge_text_feat, xe), bs,
                                     record_draws=False, grad_sync=sync)

            for i in range(10):
                flat_step(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                flat_step(args.warmup + i)
            torch.cuda.synchronize()
            ddp_proxy["flat_all_reduce"] = {"ms_per_step": (time.perf_counter() - t1) / args.steps * 1e3,
                                            "collectives_per_step": 1, "bytes": int(sync.flat.numel() * 4)}
            dist.destroy_process_group()
        except Exception as e:  # a proxy must never cost the headline line
            ddp_proxy = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        peak = 8000.0  # MI355X HBM3E spec, GB/s (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

        def k1_class(rows):
            ms, by, n = sum(r["ms"] for r in rows), sum(r["bytes"] for r in rows), len(rows)
            if n == 0 or ms <= 0:
                return None
            gbs = by / (ms * 1e-3) / 1e9
            return {"launches": n, "avg_launch_us": ms * 1e3 / n, "algorithmic_bytes_per_launch": by / n,
                    "rows_per_launch": sum(r["rows"] for r in rows) / n, "edges_per_launch": sum(r["edges"] for r in rows) / n,
                    "achieved": gbs, "frac": gbs / peak}

        k1_batch = k1_class([r for r in k1_rows if not r["augmented"]])
        k1_aug = k1_class([r for r in k1_rows if r["augmented"]])
        nb = batches[args.warmup]
        # HBM bytes from the PMC counters are collected by separate rocprofv3 --pmc passes on this workload
        # (tools/prof_pmc.sh -> profiles/step_traffic.json), NOT in this run: the file's numbers are quoted here
        tinfo = {}
        tsource = "not measured"
        if args.pmc_traffic == "auto" and world == 1 and args.workload == "c4":
            tinfo = pmc_step_traffic(args)
            tsource = tinfo.get("source") or ("live PMC pass failed (" + tinfo.get("error", "?") + ")")
        tpath = os.path.join(ROOT, "profiles", "step_traffic.json")
        if "traffic_bytes_per_step" not in tinfo and args.workload == "c4" and args.batch_size == 1024 and os.path.exists(tpath):
            with open(tpath) as fh:
                tinfo = json.load(fh)
            tsource = (tsource + "; " if "failed" in tsource else "") + \
                "quoted from profiles/step_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this " \
                "workload, tools/prof_pmc.sh; not measured in this run)"
        head = k1_batch or k1_aug or {"achieved": 0.0, "frac": 0.0}
        roofline = {"bound": "hbm", "kernel": "k_sage_agg_fwd (K1, type-indexed edge attr), launches on the batch graph "
                                              "(the teacher's two per step; rows = the nodes that can receive edges)",
                    "achieved": head["achieved"], "peak": peak, "unit": "GB/s", "frac": head["frac"],
                    "traffic": tinfo.get("k1_batch_traffic_bytes_per_launch"),
                    "traffic_source": tsource,
                    "launches": head.get("launches"), "avg_launch_us": head.get("avg_launch_us"),
                    "algorithmic_bytes_per_launch": head.get("algorithmic_bytes_per_launch"),
                    "rows_per_launch": head.get("rows_per_launch"), "edges_per_launch": head.get("edges_per_launch"),
                    "augmented_graph_launches": None if k1_aug is None else dict(
                        k1_aug, traffic=tinfo.get("k1_augmented_traffic_bytes_per_launch"),
                        note="dropout_adj(force_undirected) keeps src <= dst edges only: ~1e4 edges survive on a "
                             "sampled batch; the launch is at the launch floor")}
        out = {
            "metric": "pretrain edges/sec (fwd+bwd) on 1M-node/20M-edge synthetic graph, 1/2/4/8 GPUs",
            "value": edges / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 GEMMs (fp32 accumulation), fp32 VQ core" + (", bf16-stored features" if feat_bf16 else "")) if gemm_bf16
                     else ("f32" if not feat_bf16 else "f32 (arithmetic, VQ core, gradients) on bf16-stored features"),
            "data": "synthetic",
            "config": {"workload": wl["desc"], "nodes": wl["nodes"], "edges": wl["edges"], "feat_dim": D,
                       "layers": params["num_layers"], "vq_heads": params["codebook_head"],
                       "codebook_size": params["codebook_size"], "code_dim": params["code_dim"],
                       "seeds_per_rank": nb[3], "batch_nodes": int(nb[0].size(0)), "batch_edges": int(nb[1].num_edges),
                       "parallelism": f"dp{world}", "grad_sync": (args.grad_sync if world > 1 else None),
                       "dense_products": (
                           "bf16 GEMM mode: one bf16 MFMA pass on operands rounded to bf16, fp32 accumulation" if gemm_bf16 else
                           ("fp32 results on the 16-bit matrix cores, fp32 accumulation: big-tile core -- forward / "
                            "backward-data / code assignment from two fp16 pieces of power-of-two scaled rows (3 passes), "
                            "weight gradients from three exact bf16 pieces (6 passes)" if D >= 256 else
                            ("fp32 results on the 16-bit matrix cores, fp32 accumulation: forward / backward-data / code "
                             "assignment of the 128-column products from two fp16 pieces of power-of-two scaled rows (3 passes, "
                             "weight-stationary kernels); weight gradients and the K = 512 backward-data product from three "
                             "exact bf16 pieces (6 passes)" if ops.linear_set_pair(-1) else
                             "fp32 results from three exact bf16 pieces per operand (6 bf16 MFMA passes), fp32 accumulation: "
                             "128-row tile / weight-stationary kernels"))), "edge_attr": "type-indexed (4E + T*D*4 bytes)",
                       "preheat_steps": preheat_steps, "preheat_s": round(preheat_s, 3),
                       "loader_ms_per_batch_outside_timed_region": round(sampler_ms, 3),
                       "ms_per_step_with_loader_in_loop": None if e2e_ms is None else round(e2e_ms, 3)},
            "roofline": roofline,
            # per-step device time (HIP events around every timed step, rank 0); ms_per_step above is the wall mean
            "step_ms_median": sorted(step_ms)[len(step_ms) // 2] if step_ms else None,
            "step_ms": [round(v, 4) for v in step_ms],
        }
        # Step-level roofline: the algorithmic bytes of ONE step of this batch shape over the step time
        gs = nb[1]
        E_b = int(gs.num_edges)
        A_b = int(gs.active_rows if gs.active_rows is not None else gs.num_nodes)
        aug_edges = [r["edges"] for r in k1_rows if r["augmented"]]
        E_a = int(sum(aug_edges) / len(aug_edges)) if aug_edges else E_b
        k_s = max(int(E_b * params["topo_recon_ratio"]), 1)
        table = step_algorithmic_bytes(int(nb[0].size(0)), A_b, E_b, E_a, k_s, int(nb[3]), D,
                                       params["codebook_head"] * params["code_dim"], D, wl["types"])
        alg = float(sum(table.values()))
        step_s = dt / args.steps
        out["roofline_step"] = {"bound": "hbm", "algorithmic_bytes": alg, "peak": peak, "unit": "GB/s",
                                "achieved": alg / step_s / 1e9, "frac": alg / step_s / 1e9 / peak,
                                "floor_ms_at_peak": alg / (peak * 1e9) * 1e3,
                                "traffic": tinfo.get("traffic_bytes_per_step"),
                                "traffic_source": roofline["traffic_source"],
                                "c_node_level_passes": (alg - sum(v for k, v in table.items() if " K1 " in k)) / (nb[0].size(0) * D * 4),
                                "breakdown_MB": {k: round(v / 1e6, 1) for k, v in table.items()}}
        if bt_launches > 0 and bt_ms > 0:
            # The D >= 256 workloads are matrix-bound: their dense products and the large-codebook assignment run on the
            # big-tile core (csrc/bigtile.hip); every launch in the timed region carried its own HIP events.  EXECUTED 16-bit
            # matrix work (exact mode: three fp16 passes per fp32 product -- six bf16 passes in the weight gradients; bf16
            # mode: one) over kernel time, against the dense bf16 / fp16 MFMA peak of MI355X_MICROARCH.md (2.5 PFLOP/s;
            # what the chip sustains on random data is ~1.3-1.5).
            out["roofline_dense"] = {
                "bound": "mfma", "kernel": "k_bt_gemm (big-tile core: 256 x 256 x 64 bf16 MFMA tiles, LDS-DMA staging; "
                                           "launches of the timed region, per-launch HIP events)",
                "achieved": bt_flop / bt_ms / 1e9, "peak": 2500.0, "unit": "TFLOP/s", "frac": bt_flop / bt_ms / 1e9 / 2500.0,
                "launches_per_step": bt_launches / args.steps, "kernel_ms_per_step": bt_ms / args.steps,
                "share_of_step": bt_ms / args.steps / (dt / args.steps * 1e3),
                "executed_pflop_per_step": bt_flop / args.steps / 1e15,
                "matrix_passes_per_fp32_product": ("1 (bf16 GEMM mode)" if ops.linear_set_mode(-1) == 2 else
                                                   "3 (pair format: forward, backward-data, code assignment) / 6 (weight gradient)")}
        elif not args.no_extra:
            # The dense products are the largest share of the step: the layer product
            # lin_l(agg) + lin_r(x) at this batch's row count, timed back to back.  Executed matrix-core work is six
            # bf16 MFMA products per fp32 product (csrc/linear.hip), priced against the dense bf16 peak; informational.
            xb = batches[args.warmup][0].float()
            Mb = int(xb.size(0))
            lay = model.encoder.layers[0]
            wl_, wr_, bl_ = lay.lin_l.weight.detach(), lay.lin_r.weight.detach(), lay.lin_l.bias.detach()
            agg_ = torch.randn(A_b, D, device=dev)
            for _ in range(3):
                ops.linear_fwd(agg_, wl_, xb, wr_, bl_, True, A_b)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 30
            ev0.record()
            for _ in range(reps):
                ops.linear_fwd(agg_, wl_, xb, wr_, bl_, True, A_b)
            ev1.record()
            torch.cuda.synchronize()
            us = ev0.elapsed_time(ev1) / reps * 1e3
            flop = 2.0 * (Mb + A_b) * D * D
            x3 = ops.linear_set_mode(-1) >= 1
            pieces = 6.0 if ops.linear_set_mode(-1) == 1 else 1.0
            out["roofline_dense"] = {
                "bound": "hbm", "kernel": "k_linear_fwd_x3<128, true> (lin_l(agg[:A]) + lin_r(x) + BatchNorm statistics, one launch)",
                "rows": Mb, "aggregate_rows": A_b, "us_per_launch": us, "fp32_equivalent_tflops": flop / us / 1e6,
                "mfma_frac_of_bf16_peak": pieces * flop / us / 1e6 / (2500.0 if x3 else 157.0),
                "algorithmic_bytes": (2.0 * Mb + A_b) * D * 4, "achieved": (2.0 * Mb + A_b) * D * 4 / us / 1e3,
                "peak": peak, "unit": "GB/s", "frac": (2.0 * Mb + A_b) * D * 4 / us / 1e3 / peak}
            out["configs_extra"] = k1_at_full_graph_sizes(dev)
            if ddp_proxy is not None:
                out["configs_extra"]["dp1_under_ddp"] = ddp_proxy
        if world == 1 and not args.no_cpu_baseline:
            x, ei, xe, bs = batches[args.warmup]
            out["cpu_baseline"] = cpu_baseline(params, (x.float().cpu(), ei.edge_index.cpu(), g.edge_text_feat.cpu(), xe.cpu()), bs,
                                               args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
