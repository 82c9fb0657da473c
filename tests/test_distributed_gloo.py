"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: seed sharding, the bench's
max-time / sum-units reduction, the flat gradient averaging, DDP on the oracle model, and the
VQ EMA statistics all-reduce hook (reference vq.py:666,672)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import stem_oracle as O
    from stem_gnn_amd import parallel
    from stem_gnn_amd.model.vq import CosineSimCodebook
    torch.set_num_threads(1)
    r, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    res = {}
    # 1. seed sharding: disjoint, covering, balanced
    nodes = torch.arange(1001)
    shard = parallel.shard_seeds(nodes, rank, world, seed=7)
    pad = torch.full((501,), -1, dtype=torch.int64)
    pad[:shard.numel()] = shard
    allp = [torch.empty(501, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allp, pad)
    allv = torch.cat([a[a >= 0] for a in allp])
    res["cover"] = bool(torch.equal(torch.sort(allv).values, nodes))
    res["sizes"] = [int((a >= 0).sum()) for a in allp]
    # 2. bench reduction
    t, u = parallel.reduce_bench_stats(1.0 + rank, 100.0 * (rank + 1), torch.device("cpu"))
    res["stats"] = (t, u)
    # 3. flat gradient averaging
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 3)
    x = torch.full((2, 4), float(rank + 1))
    lin(x).sum().backward()
    local = [p.grad.clone() for p in lin.parameters()]
    parallel.allreduce_mean_grads_(lin.parameters())
    ok = True
    for p, g in zip(lin.parameters(), local):
        both = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(both, g)
        ok = ok and torch.allclose(p.grad, (both[0] + both[1]) / 2)
    res["avg_ok"] = ok
    # 3b. the step's own exchange (parallel.FlatGradSync): one fused copy, one all-reduce, p.grad become views
    torch.manual_seed(0)
    lin2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    for p in lin2[1].parameters():
        pass
    frozen = torch.nn.Parameter(torch.ones(2), requires_grad=False)
    sync = parallel.FlatGradSync(list(lin2.parameters()) + [frozen])
    ok2 = True
    for it in range(2):  # the second round finds last round's views in p.grad... replaced by fresh gradients
        for p in lin2.parameters():
            p.grad = None
        lin2(torch.full((2, 4), float(rank + 1 + it))).sum().backward()
        local = [p.grad.clone() for p in lin2.parameters()]
        sync()
        for p, g_ in zip(lin2.parameters(), local):
            both = [torch.empty_like(g_) for _ in range(world)]
            dist.all_gather(both, g_)
            ok2 = ok2 and torch.allclose(p.grad, (both[0] + both[1]) / 2) and p.grad.data_ptr() >= sync.flat.data_ptr()
    res["flat_ok"] = ok2 and len(sync.params) == 4
    # 4. DDP on the oracle pretraining model: averaged grads == grads of the mean of the two losses
    torch.manual_seed(1)
    om = O.build_oracle_model(8, 2, 2, 4, 8, ortho_max=2)
    ddp = parallel.wrap_ddp(om)
    assert all(not p.requires_grad for p in om.sem_encoder.parameters())
    g = torch.Generator().manual_seed(100 + rank)
    N, E, D, bs = 12, 30, 8, 4
    xx = torch.randn(N, D, generator=g)
    ei = torch.randint(0, N, (2, E), generator=g)
    ea = torch.randn(E, D, generator=g)
    draws = {"student_dropout": [torch.ones(N, D, dtype=torch.bool)], "teacher_dropout": [torch.ones(N, D, dtype=torch.bool)],
             "topo_perm": torch.arange(3), "neg_edge_index": torch.tensor([[0, 1, 2], [3, 4, 5]]),
             "topo_sem_perm": torch.arange(3), "ortho_ids": torch.arange(2)}
    om.train()
    _, _, _, losses = ddp((xx, ei, ea), (xx, ei, ea), bs, draws)
    params = dict(feat_lambda=100, topo_lambda=0.01, topo_sem_lambda=100, sem_lambda=1)
    O.total_loss(losses, params).backward()
    gnorm = torch.sqrt(sum((p.grad ** 2).sum() for p in om.parameters() if p.grad is not None))
    gn = [torch.zeros(()) for _ in range(world)]
    dist.all_gather(gn, gnorm)
    res["ddp_same_grads"] = bool(torch.allclose(gn[0], gn[1]))
    # 5. VQ EMA all-reduce hook arms itself when a process group exists (vq.py:771-772)
    cb = CosineSimCodebook(dim=4, codebook_size=3, num_codebooks=2, use_ddp=True, ema_update=True)
    bins = torch.full((2, 3), float(rank + 1))
    cb._all_reduce(bins)
    res["ema_allreduce"] = bins.tolist()
    out[rank] = res
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for rank in range(world):
        r = out[rank]
        assert r["cover"] and sorted(r["sizes"]) == [500, 501]
        assert r["stats"] == (2.0, 300.0)
        assert r["avg_ok"] and r["flat_ok"] and r["ddp_same_grads"]
        assert r["ema_allreduce"] == [[3.0] * 3] * 2
