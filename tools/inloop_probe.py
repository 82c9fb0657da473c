"""Where the step with the loader inside the loop spends its extra time (C4 shapes): host issue time of the step alone,
host time of the loader's three stages per batch, and the wall time of the loop under a few loader arrangements.
usage (GPU box): python tools/inloop_probe.py [steps]"""
import gc
import os
import sys
import threading
import queue
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stem_gnn_amd import ops  # noqa: E402
from stem_gnn_amd.data.sampler import HipNeighborSampler, NeighborLoader, PrefetchLoader  # noqa: E402
from stem_gnn_amd.data.synthetic import make_graph  # noqa: E402
from stem_gnn_amd.graph import EdgeTypeAttr, set_validation  # noqa: E402
from stem_gnn_amd.pretrain import build_model, build_optimizer, default_params, pretrain_step  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
D = 128
params = default_params()
params.update(input_dim=D, hidden_dim=D, code_dim=D, pretrain_batch_size=1024)
torch.manual_seed(0)
g = make_graph(1_000_000, 20_000_000, D, 4, kind="U", device=dev, graph_seed=1234, feat_seed=0)
sampler = HipNeighborSampler(g.edge_index, g.xe, g.num_nodes, g.x, g.node_text_feat, g.edge_text_feat, [10, 10], seed=100)
loader = NeighborLoader(sampler, torch.arange(g.num_nodes, device=dev), 1024, shuffle=True, seed=7)
model = build_model(params, dev)
opt, sched = build_optimizer(model, params)
set_validation(False)
model.train()


def step(b):
    pretrain_step(model, opt, sched, params, b.feat, b.graph, EdgeTypeAttr(g.edge_text_feat, b.xe), b.batch_size,
                  record_draws=False)


def prepare(b):
    b.feat = ops.gather_rows(g.node_text_feat, b.x, validate=False, capacity=b.cap_nodes)


it = iter(loader)
resident = []
for _ in range(steps + 10):
    b = next(it)
    prepare(b)
    resident.append(b)
gc.collect(); gc.freeze(); gc.disable()
pool = torch.empty(6 << 30, dtype=torch.uint8, device=dev); del pool
for b in resident[:5]:
    step(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in resident[5:]:
    step(b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"resident batches: host issue {(t1 - t0) / steps * 1e3:.3f} ms/step, wall {(t2 - t0) / steps * 1e3:.3f} ms/step")

# loader stages, host time per batch (nothing else running)
pit = loader.iter_pending()
tl = tf = tp = 0.0
for _ in range(steps):
    a = time.perf_counter(); p = next(pit); b_ = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter(); bt = p.result(); d = time.perf_counter(); prepare(bt); e = time.perf_counter()
    tl += b_ - a; tf += d - c; tp += e - d
torch.cuda.synchronize()
print(f"loader host time per batch: launch {tl / steps * 1e6:.0f} us, adopt {tf / steps * 1e6:.0f} us, gather {tp / steps * 1e6:.0f} us")


def run(name, batches):
    done, warm = 0, 5
    for b in batches:
        if done == warm:
            torch.cuda.synchronize()
            t = time.perf_counter()
            stamps = [t]
        if done == steps + warm:
            break
        step(b)
        done += 1
        if done > warm:
            stamps.append(time.perf_counter())
    done -= warm
    t_issue = time.perf_counter()
    torch.cuda.synchronize()
    per = sorted((b_ - a_) * 1e3 for a_, b_ in zip(stamps, stamps[1:]))
    print(f"{name}: {(time.perf_counter() - t) / done * 1e3:.3f} ms/step; host issue {(t_issue - t) / done * 1e3:.3f} ms/step, "
          f"per-iteration host time median {per[len(per) // 2]:.3f} p90 {per[len(per) * 9 // 10]:.3f} max {per[-1]:.3f} ms, "
          f"{sum(1 for v in per if v > 2.5)} over 2.5 ms")


class Endless:
    def iter_pending(self):
        while True:
            yield from loader.iter_pending()


class EndlessPlain:
    def __iter__(self):
        while True:
            yield from loader


run("two-deep prefetch (side stream), first pass", PrefetchLoader(Endless(), dev, prepare))
run("two-deep prefetch (side stream), second pass", PrefetchLoader(Endless(), dev, prepare))
run("one-deep prefetch (side stream)", PrefetchLoader(EndlessPlain(), dev, prepare))


def threaded(depth=3):
    q = queue.Queue(maxsize=depth)
    side = torch.cuda.Stream(device=dev)
    stop = threading.Event()

    def work():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side):
            for b in EndlessPlain():
                prepare(b)
                ev = torch.cuda.Event(); ev.record(side); b.ready = ev
                while not stop.is_set():
                    try:
                        q.put(b, timeout=0.05); break
                    except queue.Full:
                        pass
                if stop.is_set():
                    return

    th = threading.Thread(target=work, daemon=True); th.start()
    try:
        while True:
            b = q.get()
            main = torch.cuda.current_stream(dev)
            main.wait_event(b.ready)
            for v in (b.feat, b.x, b.xe, b.graph):
                v.record_stream(main)
            yield b
    finally:
        stop.set()


run("worker thread, queue of 3", threaded())
run("resident again", resident)

