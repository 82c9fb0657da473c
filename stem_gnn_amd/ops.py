"""Tensor-level wrappers over the C ABI (stem_gnn_amd/_lib.py) and the autograd Functions
built on them.  PyTorch is plumbing here: device memory, the current HIP stream and
autograd bookkeeping.  Every op requires CUDA (ROCm) tensors and raises otherwise; there
is no eager/CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Tuple

import torch
from torch import Tensor

from ._lib import EncoderCfg, GraphView, HeadsParams, SageLayer, VqParams, check, lib


_raw_stream = torch._C._cuda_getCurrentRawStream  # the hipStream_t of torch's current stream, without the Stream object
_cur_device = torch._C._cuda_getDevice


def _stream() -> int:
    return _raw_stream(_cur_device())


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()  # ctypes turns the int into the void* argument


def _req(t: Tensor, dtype, name: str, ndim: Optional[int] = None) -> Tensor:
    if not isinstance(t, Tensor):
        raise RuntimeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA (ROCm) tensor; stem_gnn_amd has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    if t.data_ptr() % 16 != 0 and t.numel() > 0:
        raise RuntimeError(f"{name}: expected 16-byte aligned storage")
    return t


def _kind(t: Tensor) -> int:
    """Element kind of a feature operand (include/stemgnn.h: STEMGNN_F32 / STEMGNN_BF16)."""
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise RuntimeError(f"feature storage must be float32 or bfloat16, got {t.dtype}")


def _workspace(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------------------
# dropout RNG keys: the keep mask is a pure function of (seed, offset, element index)
# ----------------------------------------------------------------------------------------
class _DropoutKeys(threading.local):
    def __init__(self):
        self.seed = None
        self.counter = 0


_keys = _DropoutKeys()


def manual_seed(seed: int) -> None:
    """Seed the Philox stream used by the fused dropout (and reset its call counter)."""
    _keys.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    _keys.counter = 0


def next_dropout_key() -> Tuple[int, int]:
    if _keys.seed is None:
        manual_seed(torch.initial_seed())
    _keys.counter += 1
    return _keys.seed, _keys.counter


def dropout_keep_mask(n: int, p: float, seed: int, offset: int, device) -> Tensor:
    """The boolean keep mask the fused BN/act/dropout kernels use for (seed, offset)."""
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    check(lib.stemgnn_dropout_keep_mask(n, float(p), seed, offset, _p(keep), _stream()), "dropout_keep_mask")
    return keep.bool()


# ----------------------------------------------------------------------------------------
# graph structure
# ----------------------------------------------------------------------------------------
def csr_build(edge_index: Tensor, num_nodes: int, key_row: int):
    """int64 COO [2, E] -> (rowptr [N+1], other [E], eid [E], bad_count [1]) int32, grouped by
    edge_index[key_row], stable in edge order."""
    _req(edge_index, torch.int64, "edge_index", 2)
    if edge_index.size(0) != 2:
        raise RuntimeError(f"edge_index: expected shape [2, E], got {tuple(edge_index.shape)}")
    E, N, dev = edge_index.size(1), int(num_nodes), edge_index.device
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    other = torch.empty(E, dtype=torch.int32, device=dev)
    eid = torch.empty(E, dtype=torch.int32, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    nbytes = lib.stemgnn_csr_workspace_bytes(N, E)
    ws = _workspace(nbytes, dev)
    check(lib.stemgnn_csr_build(_p(edge_index), E, N, key_row, _p(rowptr), _p(other), _p(eid), _p(bad), _p(ws),
                                ws.numel(), _stream()), "csr_build")
    return rowptr, other, eid, bad


def gather_i32(table: Tensor, index: Tensor) -> Tensor:
    _req(table, torch.int32, "table", 1)
    _req(index, torch.int32, "index", 1)
    out = torch.empty_like(index)
    check(lib.stemgnn_gather_i32(_p(table), _p(index), index.numel(), _p(out), _stream()), "gather_i32")
    return out


def graph_dropout_undirected(g, p: float, seed: int, offset: int, keep: Optional[Tensor]):
    """Both CSR views of dropout_adj(force_undirected=True) applied to graph `g` (which has both views)."""
    N, E, dev = g.num_nodes, g.num_edges, g.rowptr.device
    cap = max(2 * E, 1)
    i32 = dict(dtype=torch.int32, device=dev)
    a_rowptr = torch.empty(N + 1, **i32)
    a_src, a_eid, a_dst_t, a_eid_t = (torch.empty(cap, **i32) for _ in range(4))
    typed = g.etype_slot is not None
    if typed and g.etype_slot_t is None:
        raise RuntimeError("graph_dropout_undirected: call ensure_transpose() after set_edge_type()")
    a_type = torch.empty(cap, **i32) if typed else None
    a_type_t = torch.empty(cap, **i32) if typed else None
    inv_deg = torch.empty(N, dtype=torch.float32, device=dev)
    keep_u8 = None
    if keep is not None:
        if keep.numel() != E:
            raise RuntimeError("keep: one flag per edge expected")
        keep_u8 = keep.to(torch.uint8).contiguous()
    ws = _workspace(lib.stemgnn_graph_dropout_workspace_bytes(N), dev)
    # rows >= active_rows have no in-edges, and a survivor has source <= target: only rows below it can keep an edge
    active = -1 if g.active_rows is None else int(g.active_rows)
    check(lib.stemgnn_graph_dropout_undirected_rows(
        _p(g.rowptr), _p(g.src), _p(g.eid), _p(g.etype_slot), _p(g.rowptr_t), _p(g.dst_t), _p(g.eid_t),
        _p(g.etype_slot_t), N, E, active, float(p), seed, offset, _p(keep_u8), _p(a_rowptr), _p(a_src), _p(a_eid),
        _p(a_type), _p(a_dst_t), _p(a_eid_t), _p(a_type_t), _p(inv_deg), _p(ws), ws.numel(), _stream()),
        "graph_dropout_undirected")
    return a_rowptr, a_src, a_eid, a_type, a_dst_t, a_eid_t, a_type_t, inv_deg


def sample_subset(n: int, k: int, device, key=None) -> Tensor:
    """k distinct ids of [0, n) (what randperm(n)[:k] is used for), one kernel, no sort."""
    seed, offset = key if key is not None else next_dropout_key()
    out = torch.empty(k, dtype=torch.int64, device=device)
    check(lib.stemgnn_sample_subset(n, k, seed, offset, _p(out), _stream()), "sample_subset")
    return out


def sample_edges(edge_index: Tensor, edge_type: Optional[Tensor], k: int, want_selected: bool = False,
                 pad_columns: int = 0, key=None):
    """The picks of ``sample_subset(E, k)`` applied in the same launch: -> (perm [k], picked columns as a
    [2, k + pad_columns] buffer whose first k columns are filled, picked edge types or None, uint8 membership mask
    over the E edges or None)."""
    _req(edge_index, torch.int64, "edge_index", 2)
    E, dev = edge_index.size(1), edge_index.device
    seed, offset = key if key is not None else next_dropout_key()
    perm = torch.empty(k, dtype=torch.int64, device=dev)
    width = k + pad_columns
    sel = torch.empty(2, width, dtype=torch.int64, device=dev)
    sel_type = None
    if edge_type is not None:
        _req(edge_type, torch.int64, "edge_type", 1)
        sel_type = torch.empty(k, dtype=torch.int64, device=dev)
    selected = torch.empty(E, dtype=torch.uint8, device=dev) if want_selected else None
    check(lib.stemgnn_sample_edges(_p(edge_index), _p(edge_type), E, k, seed, offset, _p(perm), _p(sel), width,
                                   _p(sel_type), _p(selected), _stream()), "sample_edges")
    return perm, sel, sel_type, selected


def negative_sample_into(g, selected: Tensor, k: int, seed: int, offset: int, out: Tensor, column: int) -> None:
    """negative_sample written into columns [column, column + k) of the int64 [2, W] buffer `out`."""
    _req(selected, torch.uint8, "selected", 1)
    _req(out, torch.int64, "out", 2)
    if column < 0 or column + k > out.size(1):
        raise RuntimeError("negative_sample_into: column range outside the buffer")
    check(lib.stemgnn_negative_sample_into(_p(g.rowptr), _p(g.src), _p(g.eid), _p(selected), g.num_nodes, k, seed, offset,
                                           out.data_ptr() + 8 * column, out.size(1), _stream()), "negative_sample_into")


def mask_columns(x: Tensor, p: float, key=None):
    """mask_feature(x, p, mode='col'): -> (masked copy, (seed, offset)); keep mask =
    dropout_keep_mask(D, p, seed, offset).  x fp32 or bf16 (bf16 feature storage); the copy has x's dtype."""
    _req(x, x.dtype if x.dtype == torch.bfloat16 else torch.float32, "x", 2)
    seed, offset = key if key is not None else next_dropout_key()
    out = torch.empty_like(x)
    check(lib.stemgnn_mask_columns_k(_p(x), _kind(x), x.size(0), x.size(1), float(p), seed, offset, _p(out), _stream()),
          "mask_columns")
    return out, (seed, offset)


def negative_sample(g, selected: Tensor, k: int, seed: int, offset: int) -> Tensor:
    """k negative pairs for the positives flagged in `selected` (uint8 per edge of graph g)."""
    _req(selected, torch.uint8, "selected", 1)
    out = torch.empty(2, k, dtype=torch.int64, device=selected.device)
    check(lib.stemgnn_negative_sample(_p(g.rowptr), _p(g.src), _p(g.eid), _p(selected), g.num_nodes, k, seed, offset,
                                      _p(out), _stream()), "negative_sample")
    return out


def sampler_init_map(num_nodes: int, device) -> Tensor:
    local_of = torch.empty(num_nodes, dtype=torch.int32, device=device)
    check(lib.stemgnn_sampler_init_map(_p(local_of), num_nodes, _stream()), "sampler_init_map")
    return local_of


def sample_batch(rowptr: Tensor, src: Tensor, etype: Optional[Tensor], num_nodes: int, seeds: Tensor, fanouts,
                 seed: int, offset: int, local_of: Tensor):
    """One neighbour-sampled mini-batch on the device -> (n_id, b_rowptr, b_src, b_type, coo, N_b, E_b, A_b); A_b =
    leading nodes that were expanded (rows >= A_b have no in-edges).  One 12-byte device->host copy per batch."""
    _req(seeds, torch.int64, "seeds", 1)
    B, L, dev = seeds.numel(), len(fanouts), seeds.device
    level, cap_nodes, cap_edges = B, B, 0
    for f in fanouts:
        level *= int(f)
        cap_nodes += level
        cap_edges += level
    i32 = dict(dtype=torch.int32, device=dev)
    n_id = torch.empty(cap_nodes, **i32)
    b_rowptr = torch.empty(cap_nodes + 1, **i32)
    b_src = torch.empty(cap_edges, **i32)
    b_type = torch.empty(cap_edges, **i32)
    coo = torch.empty(2, cap_edges, dtype=torch.int64, device=dev)
    counts = torch.empty(3, **i32)
    ws = _workspace(lib.stemgnn_sampler_workspace_bytes(B, L, max(fanouts)), dev)
    fan = (ctypes.c_int32 * L)(*[int(f) for f in fanouts])
    check(lib.stemgnn_sample_batch(_p(rowptr), _p(src), _p(etype), num_nodes, _p(seeds), B, fan, L, seed, offset,
                                   _p(local_of), cap_nodes, cap_edges, _p(n_id), _p(b_rowptr), _p(b_src), _p(b_type),
                                   _p(coo), _p(counts), _p(ws), ws.numel(), _stream()), "sample_batch")
    nb, eb, ab = counts.tolist()
    return n_id[:nb], b_rowptr[:nb + 1], b_src[:eb], b_type[:eb], coo[:, :eb].contiguous(), nb, eb, ab


class _SamplerPlan:
    """Where every output of stemgnn_sample_batch_views lives inside ONE allocation, for a seed count and fan-outs (the
    capacities depend on nothing else): byte offsets, 256-aligned, int64 parts first, the kernels' workspace last."""
    _cache: dict = {}

    def __init__(self, B: int, fanouts):
        level, cn, ce = B, B, 0
        for f in fanouts:
            level *= int(f)
            cn += level
            ce += level
        self.B, self.L, self.cn, self.ce = B, len(fanouts), cn, ce
        self.fan = (ctypes.c_int32 * self.L)(*[int(f) for f in fanouts])
        self.ws_bytes = int(lib.stemgnn_sampler_workspace_bytes(B, self.L, max(fanouts)))
        self.off, at = {}, 0
        for name, nbytes in (("coo", 16 * ce), ("n_id64", 8 * cn), ("x", 8 * cn), ("type64", 8 * ce), ("n_id", 4 * cn),
                             ("rowptr", 4 * cn + 4), ("src", 4 * ce), ("type", 4 * ce), ("rowptr_t", 4 * cn + 4),
                             ("dst_t", 4 * ce), ("eid_t", 4 * ce), ("type_t", 4 * ce), ("inv_deg", 4 * cn),
                             ("ws", self.ws_bytes)):
            self.off[name] = at
            at += (nbytes + 255) // 256 * 256
        self.total = at

    @classmethod
    def of(cls, B: int, fanouts) -> "_SamplerPlan":
        key = (B, tuple(int(f) for f in fanouts))
        plan = cls._cache.get(key)
        if plan is None:
            plan = cls._cache[key] = cls(B, fanouts)
        return plan


class PendingBatchViews:
    """A batch whose sampler launches are enqueued and whose sizes are on their way to the host: ``result()`` waits for
    the 12 bytes of sizes only (nothing to wait for when the batch was launched a step ahead) and cuts the views."""

    def __init__(self, plan: _SamplerPlan, slab: Tensor, counts_host: Tensor, event, by_source: bool = True):
        self.plan, self.slab, self.counts_host, self.event, self.by_source = plan, slab, counts_host, event, by_source

    def result(self) -> dict:
        self.event.synchronize()
        nb, eb, ab = self.counts_host.tolist()
        p, off = self.plan, self.plan.off
        s32, s64, f32 = self.slab.view(torch.int32), self.slab.view(torch.int64), self.slab.view(torch.float32)
        i32 = lambda name, n: s32[off[name] >> 2:(off[name] >> 2) + n]
        i64 = lambda name, n: s64[off[name] >> 3:(off[name] >> 3) + n]
        out = dict(nb=nb, eb=eb, ab=ab, cap_nodes=p.cn, cap_edges=p.ce, slab=self.slab, n_id=i32("n_id", nb),
                   rowptr=i32("rowptr", nb + 1), src=i32("src", eb), type=i32("type", eb),
                   coo=i64("coo", 2 * eb).view(2, eb), rowptr_t=None,
                   inv_deg=f32[off["inv_deg"] >> 2:(off["inv_deg"] >> 2) + nb], n_id64=i64("n_id64", nb),
                   x=i64("x", nb), type64=i64("type64", eb))
        if self.by_source:
            out.update(rowptr_t=i32("rowptr_t", nb + 1), dst_t=i32("dst_t", eb), eid_t=i32("eid_t", eb),
                       type_t=i32("type_t", eb))
        return out


def sample_batch_views_launch(rowptr: Tensor, src: Tensor, etype: Optional[Tensor], num_nodes: int, seeds: Tensor,
                              fanouts, seed: int, offset: int, local_of: Tensor, x: Optional[Tensor] = None,
                              counts_host: Optional[Tensor] = None, by_source: bool = True) -> PendingBatchViews:
    """sample_batch plus the by-source CSR, 1 / in-degree and the int64 forms of the batch (n_id, edge types, feature
    rows x[n_id]) from the same ten-odd launches; ONE allocation holds every output and the kernels' workspace.  The
    kernels write the sizes straight into ``counts_host`` (pinned int32 [3], device-accessible host memory); nothing
    here waits for the device.  ``by_source=False`` leaves the by-source CSR out (graphs with long out-rows: its
    ordering pass walks a row in one thread); ``GraphStructure.ensure_transpose`` then sorts when somebody asks."""
    _req(seeds, torch.int64, "seeds", 1)
    if x is not None:
        _req(x, torch.int64, "x", 1)
    plan = _SamplerPlan.of(seeds.numel(), fanouts)
    slab = torch.empty(plan.total, dtype=torch.uint8, device=seeds.device)
    if counts_host is None:
        counts_host = torch.empty(3, dtype=torch.int32, pin_memory=True)
    base, off = slab.data_ptr(), plan.off
    check(lib.stemgnn_sample_batch_views(
        _p(rowptr), _p(src), _p(etype), num_nodes, _p(seeds), plan.B, plan.fan, plan.L, seed, offset, _p(local_of),
        plan.cn, plan.ce, base + off["n_id"], base + off["rowptr"], base + off["src"], base + off["type"],
        base + off["coo"], _p(counts_host), *((base + off["rowptr_t"], base + off["dst_t"], base + off["eid_t"],
                                               base + off["type_t"]) if by_source else (None,) * 4),
        base + off["inv_deg"], base + off["n_id64"], base + off["type64"], _p(x), base + off["x"],
        base + off["ws"], plan.ws_bytes, _stream()), "sample_batch_views")
    event = torch.cuda.Event()
    event.record()
    return PendingBatchViews(plan, slab, counts_host, event, by_source)


def sample_batch_views(*args, **kwargs) -> dict:
    return sample_batch_views_launch(*args, **kwargs).result()


def _sample_batch_full(rowptr: Tensor, src: Tensor, etype: Optional[Tensor], num_nodes: int, seeds: Tensor, fanouts,
                      seed: int, offset: int, local_of: Tensor, x: Optional[Tensor] = None) -> dict:
    """A batch whose fan-outs include -1 (every in-neighbour; the reference's evaluation loaders): built hop by hop, the
    host reading one size per hop to allocate the hop's entries (stemgnn_sampler_full_*).  Same keys as
    ``sample_batch_views(...)``, without the by-source view (an evaluation batch has no backward; ``ensure_transpose``
    sorts on demand)."""
    _req(seeds, torch.int64, "seeds", 1)
    if x is not None:
        _req(x, torch.int64, "x", 1)
    B, L, dev, st = seeds.numel(), len(fanouts), seeds.device, _stream()
    i32 = dict(dtype=torch.int32, device=dev)
    state = torch.empty(32, **i32)
    size = torch.empty(1, dtype=torch.int32, pin_memory=True)
    n_id = torch.empty(B, **i32)
    check(lib.stemgnn_sampler_full_begin(_p(seeds), B, num_nodes, _p(local_of), _p(n_id), _p(state), st), "sampler_full_begin")
    cap, known, hops = B, B, []
    for h, f in enumerate(int(f) for f in fanouts):
        cnt, ent, wins, base = (torch.empty(cap, **i32) for _ in range(4))
        check(lib.stemgnn_sampler_full_hop_sizes(_p(rowptr), _p(n_id), _p(state), h, f, cap, _p(cnt), _p(ent), _p(size), st),
              "sampler_full_hop_sizes")
        torch.cuda.current_stream(dev).synchronize()
        total = int(size[0])
        s_src, s_type = torch.empty(max(total, 1), **i32), torch.empty(max(total, 1), **i32)
        n_cap = known + min(total, num_nodes)
        if n_cap > n_id.numel():
            grown = torch.empty(n_cap, **i32)
            grown[:known] = n_id[:known]
            n_id = grown
        check(lib.stemgnn_sampler_full_hop_expand(_p(rowptr), _p(src), _p(etype), _p(n_id), n_cap, _p(state), h, f, seed,
                                                  offset, B, cap, total, _p(cnt), _p(ent), _p(s_src), _p(s_type), _p(wins),
                                                  _p(base), _p(local_of), None, st), "sampler_full_hop_expand")
        hops.append((cap, cnt, ent, s_src, s_type, wins, base))
        cap, known = max(min(total, num_nodes), 1), n_cap
    counts = state.tolist()  # nodes[0..15] | edges[0..15]: the one read the outputs' sizes need
    nb, eb, ab = counts[L + 1], counts[16 + L], min(counts[L], counts[L + 1])
    rp, inv = torch.empty(nb + 1, **i32), torch.empty(nb, dtype=torch.float32, device=dev)
    b_src, b_type = torch.empty(eb, **i32), torch.empty(eb, **i32)
    even = lambda n: n + (n & 1)  # every part starts on a 16-byte boundary (what the kernels' vector loads want)
    o1, o2, o3 = 2 * eb, 2 * eb + even(nb), 2 * eb + 2 * even(nb)
    s64 = torch.empty(o3 + eb, dtype=torch.int64, device=dev)
    coo, n_id64, x_out, type64 = s64[:2 * eb], s64[o1:o1 + nb], s64[o2:o2 + nb], s64[o3:o3 + eb]
    ws = _workspace(lib.stemgnn_sampler_full_finish_workspace_bytes(eb), dev)
    sizes = torch.empty(3, **i32)  # (N_b, E_b, A_b) once more, for callers of the C entry point; known here already
    ptrs = lambda k: (ctypes.c_void_p * L)(*[_p(hp[k]) for hp in hops])
    check(lib.stemgnn_sampler_full_finish(
        _p(state), L, (ctypes.c_int32 * L)(*[int(f) for f in fanouts]), (ctypes.c_int64 * L)(*[hp[0] for hp in hops]),
        ptrs(1), ptrs(2), ptrs(3), ptrs(4), _p(local_of), _p(n_id), nb, eb, _p(rp), _p(b_src), _p(b_type), _p(coo), _p(inv),
        _p(n_id64), _p(type64), _p(x), _p(x_out), _p(sizes), _p(ws), ws.numel(), st), "sampler_full_finish")
    return dict(nb=nb, eb=eb, ab=ab, cap_nodes=nb, cap_edges=eb, slab=s64, n_id=n_id[:nb], rowptr=rp, src=b_src, type=b_type,
                coo=coo.view(2, eb), rowptr_t=None, inv_deg=inv, n_id64=n_id64, x=x_out, type64=type64)


def sample_batch_full(rowptr: Tensor, src: Tensor, etype: Optional[Tensor], num_nodes: int, seeds: Tensor, fanouts,
                      seed: int, offset: int, local_of: Tensor, x: Optional[Tensor] = None) -> dict:
    """``_sample_batch_full`` with the scratch map restored if a step fails half-way (an allocation that does not fit,
    a bad argument): claimed-but-unnumbered entries would otherwise corrupt every later batch of the sampler."""
    try:
        return _sample_batch_full(rowptr, src, etype, num_nodes, seeds, fanouts, seed, offset, local_of, x)
    except BaseException:
        check(lib.stemgnn_sampler_init_map(_p(local_of), local_of.numel(), _stream()), "sampler_init_map")
        raise


def inv_degree(rowptr: Tensor) -> Tensor:
    _req(rowptr, torch.int32, "rowptr", 1)
    n = rowptr.numel() - 1
    out = torch.empty(n, dtype=torch.float32, device=rowptr.device)
    check(lib.stemgnn_inv_degree(_p(rowptr), n, _p(out), _stream()), "inv_degree")
    return out


# ----------------------------------------------------------------------------------------
# K1 / K2
# ----------------------------------------------------------------------------------------
class KernelTimer:
    """In-situ timing of every K1-forward launch (bench.py's roofline leg).  The library stamps
    each launch with its own begin/end HIP events (hipExtLaunchKernelGGL), so the time is the
    kernel's, not the gap-inclusive span between marker packets.  ``bytes`` lists, in launch order, what each launch
    worked on: (rows computed, graph, D, edge mode, T)."""

    def __init__(self):
        self.enabled = False
        self.bytes = []

    def reset(self, enabled: bool):
        self.enabled = enabled
        self.bytes = []
        check(lib.stemgnn_profile_k1(1 if enabled else 0), "profile_k1")

    def collect(self):
        """-> (total kernel ms, launches, total algorithmic bytes)"""
        rows = self.collect_each()
        return sum(r["ms"] for r in rows), len(rows), float(sum(r["bytes"] for r in rows))

    def collect_each(self):
        """-> one dict per launch, in launch order: ms, algorithmic bytes, rows, live edges, augmented (bool)."""
        cap = max(len(self.bytes), 1) + 64
        ms = (ctypes.c_float * cap)()
        n = ctypes.c_int64(0)
        check(lib.stemgnn_profile_k1_collect_each(ms, cap, ctypes.byref(n)), "profile_k1_collect_each")
        out = []
        for i, (N, g, D, mode, T) in enumerate(self.bytes[:n.value]):
            E = g.live_edges_host()
            out.append(dict(ms=float(ms[i]), bytes=k1_algorithmic_bytes(N, E, D, mode, T), rows=N, edges=E,
                            augmented=g._edge_index is None))
        return out


k1_timer = KernelTimer()


def k1_algorithmic_bytes(N: int, E: int, D: int, mode: str, T: int = 0) -> int:
    """E*D*4 (source rows) + A + 4E (src ids) + 4(N+1) (rowptr) + N*D*4 (output);
    A = E*D*4 + 4E (dense rows + edge ids) | 4E + T*D*4 (type ids + table) | 0.  N = the rows the launch computes
    and writes (a sampled batch: only the rows that can receive edges)."""
    a = {"dense": E * D * 4 + 4 * E, "table": 4 * E + T * D * 4, "none": 0}[mode]
    return E * D * 4 + a + 4 * E + 4 * (N + 1) + N * D * 4


def sage_agg_fwd(x: Tensor, graph, edge_attr: Optional[Tensor], etab: Optional[Tensor]) -> Tensor:
    _req(x, torch.float32, "x", 2)
    N, D = x.shape
    if N != graph.num_nodes:
        raise RuntimeError(f"x has {N} rows but the graph structure was built for {graph.num_nodes} nodes")
    etype_slot = None
    T = 0
    if graph.num_edges == 0:
        edge_attr = etab = None  # no edges: the edge term is never read
    if edge_attr is not None:
        _req(edge_attr, torch.float32, "edge_attr", 2)
        if tuple(edge_attr.shape) != (graph.num_edges, D):
            raise RuntimeError(f"edge_attr: expected shape {(graph.num_edges, D)}, got {tuple(edge_attr.shape)}")
    if etab is not None:
        _req(etab, torch.float32, "edge_type_table", 2)
        if etab.size(1) != D:
            raise RuntimeError("edge_type_table: feature dim mismatch")
        T = etab.size(0)
        etype_slot = graph.etype_slot
        if etype_slot is None:
            raise RuntimeError("graph structure has no edge types; build it with edge_type=...")
    agg = torch.empty_like(x)
    plan = graph.split_plan("in")
    if plan is None:
        check(lib.stemgnn_sage_agg_fwd(_p(x), N, D, _p(graph.rowptr), _p(graph.src), _p(graph.eid), _p(edge_attr),
                                       _p(etab), _p(etype_slot), T, _p(agg), _stream()), "sage_agg_fwd")
    else:
        check(lib.stemgnn_sage_agg_fwd_split(_p(x), N, D, graph.slot_capacity, _p(graph.rowptr), _p(graph.src),
                                             _p(graph.eid), _p(edge_attr), _p(etab), _p(etype_slot), T, _p(agg), 1,
                                             *plan.args(D, x.device), _stream()), "sage_agg_fwd_split")
    if k1_timer.enabled:
        mode = "dense" if edge_attr is not None else ("table" if etab is not None else "none")
        k1_timer.bytes.append((N, graph, D, mode, T))  # edge counts are resolved after the timed region
    return agg


def sage_agg_bwd(g_agg: Tensor, x: Tensor, graph, edge_attr: Optional[Tensor], etab: Optional[Tensor]) -> Tensor:
    _req(g_agg, torch.float32, "g_agg", 2)
    N, D = x.shape
    graph.ensure_transpose()
    if graph.num_edges == 0:
        edge_attr = etab = None
    T = 0 if etab is None else etab.size(0)
    g_x = torch.empty_like(x)
    ets = graph.etype_slot_t if etab is not None else None
    plan = graph.split_plan("out")
    if plan is None:
        check(lib.stemgnn_sage_agg_bwd(_p(g_agg), _p(x), N, D, _p(graph.rowptr_t), _p(graph.dst_t), _p(graph.eid_t),
                                       _p(graph.inv_deg), _p(edge_attr), _p(etab), _p(ets), T, _p(g_x), _stream()),
              "sage_agg_bwd")
    else:
        check(lib.stemgnn_sage_agg_bwd_split(_p(g_agg), _p(x), N, D, graph.slot_capacity, _p(graph.rowptr_t),
                                             _p(graph.dst_t), _p(graph.eid_t), _p(graph.inv_deg), _p(edge_attr),
                                             _p(etab), _p(ets), T, _p(g_x), 1, *plan.args(D, x.device), _stream()),
              "sage_agg_bwd_split")
    return g_x


SPLIT_CHUNK = 64    # edges per work item of a heavy row
SPLIT_HEAVY = 128   # rows with more edges than this are split


class SplitPlan:
    """Device buffers of the heavy-row split plan of one CSR (include/stemgnn.h: stemgnn_sage_agg_fwd_split).
    The first aggregation call over the CSR fills it on the device; later calls reuse it."""

    def __init__(self, num_edges: int, device):
        """``num_edges``: an upper bound of the CSR's LIVE slots (GraphStructure.slot_capacity)."""
        self.chunk, self.heavy = SPLIT_CHUNK, max(SPLIT_HEAVY, SPLIT_CHUNK)
        self.cap_items = 2 * (num_edges // self.chunk) + 2
        self.cap_heavy = num_edges // self.chunk + 1
        i32 = dict(dtype=torch.int32, device=device)
        self.item_row = torch.empty(self.cap_items, **i32)
        self.item_beg = torch.empty(self.cap_items, **i32)
        self.heavy_row = torch.empty(self.cap_heavy, **i32)
        self.heavy_span = torch.empty(self.cap_heavy, 2, **i32)
        self.counts = torch.zeros(2, **i32)
        self.built = False

    def args(self, D: int, device):
        partial = torch.empty(self.cap_items, D, dtype=torch.float32, device=device)
        build = 0 if self.built else 1
        self.built = True
        return (self.chunk, self.heavy, build, self.cap_items, self.cap_heavy, _p(self.item_row), _p(self.item_beg),
                _p(self.heavy_row), _p(self.heavy_span), _p(self.counts), _p(partial))


class SageAggFn(torch.autograd.Function):
    """agg = mean_{j->i} relu(x_j + ea_ji)  (reference model/encoder.py:82,94-97)."""

    @staticmethod
    def forward(ctx, x, graph, edge_attr, etab):
        x = x.contiguous()
        agg = sage_agg_fwd(x, graph, edge_attr, etab)
        ctx.graph = graph
        ctx.save_for_backward(x, edge_attr, etab)
        return agg

    @staticmethod
    def backward(ctx, g_agg):
        x, edge_attr, etab = ctx.saved_tensors
        g_x = None
        if ctx.needs_input_grad[0]:
            g_x = sage_agg_bwd(g_agg.contiguous(), x, ctx.graph, edge_attr, etab)
        return g_x, None, None, None


class MeanAggFn(torch.autograd.Function):
    """agg[i] = mean_{j -> i} x_j without edge term or relu: scatter_mean(x[col], row) of
    MixtureSageLayer (reference model/encoder.py:124) on the CSR of the flipped graph."""

    @staticmethod
    def forward(ctx, x, graph):
        x = x.contiguous()
        _req(x, torch.float32, "x", 2)
        N, D = x.shape
        if N != graph.num_nodes:
            raise RuntimeError("x / graph size mismatch")
        agg = torch.empty_like(x)
        plan = graph.split_plan("in")
        if plan is None:
            check(lib.stemgnn_mean_agg_fwd(_p(x), N, D, _p(graph.rowptr), _p(graph.src), _p(agg), _stream()),
                  "mean_agg_fwd")
        else:
            check(lib.stemgnn_sage_agg_fwd_split(_p(x), N, D, graph.slot_capacity, _p(graph.rowptr), _p(graph.src), None,
                                                 None, None, None, 0, _p(agg), 0, *plan.args(D, x.device), _stream()),
                  "mean_agg_fwd_split")
        ctx.graph = graph
        return agg

    @staticmethod
    def backward(ctx, g_agg):
        g = ctx.graph
        g.ensure_transpose()
        g_agg = g_agg.contiguous()
        N, D = g_agg.shape
        g_x = torch.empty_like(g_agg)
        plan = g.split_plan("out")
        if plan is None:
            check(lib.stemgnn_mean_agg_bwd(_p(g_agg), N, D, _p(g.rowptr_t), _p(g.dst_t), _p(g.inv_deg), _p(g_x),
                                           _stream()), "mean_agg_bwd")
        else:
            check(lib.stemgnn_sage_agg_bwd_split(_p(g_agg), None, N, D, g.slot_capacity, _p(g.rowptr_t), _p(g.dst_t), None,
                                                 _p(g.inv_deg), None, None, None, 0, _p(g_x), 0,
                                                 *plan.args(D, g_agg.device), _stream()), "mean_agg_bwd_split")
        return g_x, None


# ----------------------------------------------------------------------------------------
# Encoder phase: the whole layer stack in one call per direction (csrc/phases.hip)
# ----------------------------------------------------------------------------------------
def graph_view(graph, need_transpose: bool) -> GraphView:
    """stemgnn_graph_view of a GraphStructure (pointers only; the structure owns the memory)."""
    if need_transpose:
        graph.ensure_transpose()
    gv = GraphView()
    gv.num_nodes = graph.num_nodes
    gv.active_rows = graph.num_nodes if graph.active_rows is None else graph.active_rows
    gv.rowptr, gv.src, gv.eid, gv.etype_slot = _p(graph.rowptr), _p(graph.src), _p(graph.eid), _p(graph.etype_slot)
    gv.rowptr_t, gv.dst_t, gv.eid_t = _p(graph.rowptr_t), _p(graph.dst_t), _p(graph.eid_t)
    gv.etype_slot_t, gv.inv_deg = _p(graph.etype_slot_t), _p(graph.inv_deg)
    return gv


def encoder_phase_ok(graph, x: Tensor, dense: Optional[Tensor], etab: Optional[Tensor]) -> bool:
    """The fused phase covers what the pretraining path runs: fp32 CUDA rows, a graph without heavy rows (no split
    plan on either side) and at most one edge-attribute form."""
    if not (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.dim() == 2 and x.size(1) % 4 == 0):
        return False
    if graph.num_nodes != x.size(0) or graph.num_nodes == 0:
        return False
    if dense is not None and (dense.requires_grad or tuple(dense.shape) != (graph.num_edges, x.size(1))):
        return False
    if etab is not None and (graph.etype_slot is None or etab.size(1) != x.size(1)):
        return False
    if dense is not None and etab is not None:
        return False
    return graph.split_plan("in") is None and graph.split_plan("out") is None


class EncoderFn(torch.autograd.Function):
    """Encoder.forward for the 'sage' backbone without MoE layers (reference model/encoder.py:279-323) as ONE library
    call, and its backward as one more.  ``meta`` = (cfg dict, per-layer dicts with the BatchNorm buffers and the
    dropout key); ``params`` = per layer (lin_l.weight, lin_l.bias or None, lin_r.weight, bn.weight or None,
    bn.bias or None)."""

    @staticmethod
    def _layers(meta, params, grads=None):
        cfg_d, per_layer = meta
        L = len(per_layer)
        arr = (SageLayer * L)()
        for l, info in enumerate(per_layer):
            w_l, b_l, w_r, bn_w, bn_b = params[5 * l:5 * l + 5]
            y = arr[l]
            y.out_dim, y.in_dim = w_l.shape
            y.w_l, y.b_l, y.w_r, y.bn_weight, y.bn_bias = _p(w_l), _p(b_l), _p(w_r), _p(bn_w), _p(bn_b)
            y.bn_running_mean, y.bn_running_var = _p(info["running_mean"]), _p(info["running_var"])
            y.bn_num_batches_tracked = _p(info["num_batches_tracked"])
            y.bn_eps, y.bn_momentum = info["eps"], info["momentum"]
            y.drop_seed, y.drop_offset = info["drop_key"]
            if grads is not None:
                g = grads[5 * l:5 * l + 5]
                y.g_w_l, y.g_b_l, y.g_w_r, y.g_bn_weight, y.g_bn_bias = (_p(t) for t in g)
        cfg = EncoderCfg(L, int(cfg_d["use_bn"]), int(cfg_d["training"]), int(cfg_d["act"]), float(cfg_d["slope"]),
                         float(cfg_d["p"]), int(cfg_d.get("out_rows") or 0), int(cfg_d.get("feature_kind") or 0))
        return arr, cfg

    @staticmethod
    def forward(ctx, x, graph, dense, etab, meta, *params):
        x = x.contiguous()
        for t in params:
            if t is not None:
                _req(t, torch.float32, "encoder parameter")
        meta = (dict(meta[0], feature_kind=_kind(x)), meta[1])  # the input's dtype IS the storage mode
        if x.dtype == torch.bfloat16 and x.requires_grad:
            raise RuntimeError("encoder phase: bf16-stored features take no gradient")
        arr, cfg = EncoderFn._layers(meta, params)
        need_bwd = bool(meta[0].get("wants_grad", True))  # grad mode is off inside forward: the caller decides
        gv = graph_view(graph, need_bwd)
        N, A = gv.num_nodes, gv.active_rows
        nbytes = lib.stemgnn_encoder_save_bytes(N, A, arr, ctypes.byref(cfg))
        if nbytes == 0:
            raise RuntimeError("encoder phase: unsupported layer configuration")
        save = _workspace(nbytes, x.device)
        rows = cfg.out_rows if 0 < cfg.out_rows < N else N
        if rows < N and need_bwd:
            raise RuntimeError("encoder phase: out_rows is a forward-only option (no_grad callers)")
        z = torch.empty(rows, arr[len(arr) - 1].out_dim, dtype=torch.float32, device=x.device)
        T = 0 if etab is None else etab.size(0)
        linear_scratch(N, 2 * max(int(a.in_dim) for a in arr), max(int(a.out_dim) for a in arr))
        check(lib.stemgnn_encoder_fwd(ctypes.byref(gv), _p(x), _p(dense), _p(etab), T, arr, ctypes.byref(cfg), _p(z),
                                      _p(save), save.numel(), _stream()), "encoder_fwd")
        if k1_timer.enabled and A > 0:  # one K1 launch per layer, over the rows that can receive edges
            mode = "dense" if dense is not None else ("table" if etab is not None else "none")
            for l in range(len(arr)):
                k1_timer.bytes.append((A, graph, int(arr[l].in_dim), mode, T))
        ctx.graph, ctx.meta = graph, meta
        ctx.save_for_backward(x, dense, etab, save, z, *params)
        return z

    @staticmethod
    def backward(ctx, g_z):
        x, dense, etab, save, z, *params = ctx.saved_tensors
        need = ctx.needs_input_grad
        grads = [torch.empty_like(t) if (t is not None and need[5 + i]) else None for i, t in enumerate(params)]
        # a weight gradient is produced together with its layer's bias gradient: give lin_r's product a target
        # whenever either is wanted (see stemgnn_encoder_bwd)
        for l in range(len(params) // 5):
            if grads[5 * l + 1] is not None and grads[5 * l + 2] is None:
                grads[5 * l + 2] = torch.empty_like(params[5 * l + 2])
        arr, cfg = EncoderFn._layers(ctx.meta, params, grads)
        gv = graph_view(ctx.graph, True)
        N, A = gv.num_nodes, gv.active_rows
        g_x = torch.empty_like(x) if need[0] else None
        scratch = _workspace(lib.stemgnn_encoder_bwd_scratch_bytes(N, A, arr, ctypes.byref(cfg)), x.device)
        g_work = g_z.contiguous()
        T = 0 if etab is None else etab.size(0)
        linear_scratch(N, 2 * max(int(a.in_dim) for a in arr), max(int(a.out_dim) for a in arr), factor=len(arr) + 1)
        check(lib.stemgnn_encoder_bwd(ctypes.byref(gv), _p(x), _p(dense), _p(etab), T, arr, ctypes.byref(cfg),
                                      _p(g_work), _p(g_x), _p(save), save.numel(), _p(scratch), scratch.numel(),
                                      _stream()), "encoder_bwd")
        out = [g if need[5 + i] else None for i, g in enumerate(grads)]
        return (g_x, None, None, None, None, *out)


# ----------------------------------------------------------------------------------------
# K4
# ----------------------------------------------------------------------------------------
def bn_stats(y: Tensor, eps: float, running_mean: Optional[Tensor], running_var: Optional[Tensor],
             momentum: float) -> Tuple[Tensor, Tensor]:
    _req(y, torch.float32, "y", 2)
    N, D = y.shape
    mean = torch.empty(D, dtype=torch.float32, device=y.device)
    rstd = torch.empty_like(mean)
    ws = _workspace(lib.stemgnn_bn_workspace_bytes(N, D), y.device)
    check(lib.stemgnn_bn_stats(_p(y), N, D, float(eps), _p(mean), _p(rstd), _p(running_mean), _p(running_var),
                               float(momentum), _p(ws), ws.numel(), _stream()), "bn_stats")
    return mean, rstd


def bn_stats_from_partials(partial: Tensor, blocks: int, num_rows: int, eps: float, running_mean: Optional[Tensor],
                           running_var: Optional[Tensor], momentum: float,
                           num_batches_tracked: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Finalise BatchNorm statistics from the column partials LinearFn's fused epilogue wrote (and bump the
    module's int64 call counter in the same launch when it is given)."""
    if num_batches_tracked is not None:
        _req(num_batches_tracked, torch.int64, "num_batches_tracked")
    D = partial.size(-1)
    mean = torch.empty(D, dtype=torch.float32, device=partial.device)
    rstd = torch.empty_like(mean)
    check(lib.stemgnn_bn_stats_from_partials(_p(partial), blocks, num_rows, D, float(eps), _p(mean), _p(rstd),
                                             _p(running_mean), _p(running_var), float(momentum),
                                             _p(num_batches_tracked), _stream()),
          "bn_stats_from_partials")
    return mean, rstd


class BnActDropFn(torch.autograd.Function):
    """dropout(act(batch_norm(y))) with training statistics (reference model/encoder.py:313-317).
    ``stats`` = (mean, rstd) already computed by a producer (LinearFn's fused epilogue)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, use_bn, momentum, eps, act, slope, p, seed, offset,
                stats=None):
        y = y.contiguous()
        _req(y, torch.float32, "y", 2)
        N, D = y.shape
        mean = rstd = None
        if use_bn:
            if N <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(y.shape)}")
            if stats is not None:
                mean, rstd = stats
            else:
                mean, rstd = bn_stats(y, eps, running_mean, running_var, momentum)
        out = torch.empty_like(y)
        g = gamma if use_bn else None
        b = beta if use_bn else None
        check(lib.stemgnn_bn_act_drop_fwd(_p(y), N, D, _p(mean), _p(rstd), _p(g), _p(b), int(act), float(slope),
                                          float(p), seed, offset, _p(out), _stream()), "bn_act_drop_fwd")
        ctx.save_for_backward(y, mean, rstd, g, b)
        ctx.cfg = (int(act), float(slope), float(p), seed, offset)
        return out

    @staticmethod
    def backward(ctx, g_out):
        y, mean, rstd, gamma, beta = ctx.saved_tensors
        act, slope, p, seed, offset = ctx.cfg
        N, D = y.shape
        g_out = g_out.contiguous()
        g_y = torch.empty_like(y)
        g_gamma = g_beta = None
        ws = None
        if mean is not None:
            g_gamma = torch.empty_like(gamma)
            g_beta = torch.empty_like(beta)
            ws = _workspace(lib.stemgnn_bn_workspace_bytes(N, D), y.device)
        check(lib.stemgnn_bn_act_drop_bwd(_p(g_out), _p(y), N, D, _p(mean), _p(rstd), _p(gamma), _p(beta), act, slope, p,
                                          seed, offset, _p(g_y), _p(g_gamma), _p(g_beta), _p(ws),
                                          0 if ws is None else ws.numel(), _stream()), "bn_act_drop_bwd")
        return (g_y, g_gamma, g_beta) + (None,) * 11


# ----------------------------------------------------------------------------------------
# K3 / K5: dense projections on the fp32 matrix cores
# ----------------------------------------------------------------------------------------
def linear_set_mode(mode: int) -> int:
    """1 (default): fp32 products from exact bf16 pieces on the bf16 matrix cores; 0: fp32-MFMA kernels; 2: bf16 GEMMs
    (BASELINE config 5: operands of every Linear rounded to bf16, one matrix pass, fp32 accumulation; the quantiser's
    similarity / arg-max core stays exact).  Returns the previous mode (any other argument only queries)."""
    return int(lib.stemgnn_linear_set_mode(int(mode)))


def linear_set_bigtile(on: int) -> int:
    """Large products (the D = 768 configurations) on the big-tile core (default) or on the 128-row tile kernels (0).
    Returns the previous setting (any argument but 0 / 1 only queries)."""
    return int(lib.stemgnn_linear_set_bigtile(int(on)))


def linear_set_pair(on: int) -> int:
    """Exact mode on the big-tile core: forward / backward-data / code assignment from two fp16 pieces of power-of-two
    scaled rows (1, default: three matrix passes) or from the three bf16 pieces (0: six).  Returns the previous setting."""
    return int(lib.stemgnn_linear_set_pair(int(on)))


# The big-tile core's scratch (operand planes, split slabs, arg-max candidates) is the CALLER's: one arena per (device,
# stream), allocated here through torch's caching allocator -- in stream order, before the call that needs it -- and
# registered with the library, which never allocates.  It is kept between calls and only ever grows.
_BT_ARENA = {}


def linear_scratch(rows: int, dim_a: int, dim_b: int, vq: Optional[Tuple[int, int, int]] = None, factor: int = 1) -> None:
    """Make sure the current (device, stream)'s arena covers every product with at most ``rows`` rows and feature extents
    ``dim_a`` x ``dim_b`` (forward, backward-data, weight gradient) and, with ``vq = (heads, code_dim, codebook_size)``,
    the quantiser's large-codebook assignment over ``rows`` rows.  ``factor``: a backward phase keeps the planes of every
    operand it has cut until it ends (they are shared between its products), so it asks for a multiple of the
    single-product bound.  Shapes the big-tile core does not take cost nothing."""
    need = 0
    if rows >= 8192 and min(dim_a, dim_b) >= 256:
        need = lib.stemgnn_linear_scratch_bytes(rows, dim_a, dim_b) * max(int(factor), 1)
    if vq is not None and rows >= 8192 and vq[2] >= 512 and vq[1] >= 256:
        need = max(need, lib.stemgnn_vq_assign_scratch_bytes(rows, vq[0], vq[1], vq[2]))
    if need == 0:
        return
    dev, st = _cur_device(), _stream()
    cur = _BT_ARENA.get((dev, st))
    if cur is None or cur.numel() < need:
        t = torch.empty(need + need // 16, dtype=torch.uint8, device=torch.device("cuda", dev))
        check(lib.stemgnn_linear_set_scratch(_p(t), t.numel(), st), "linear_set_scratch")
        _BT_ARENA[(dev, st)] = t  # the old block goes back to the allocator in stream order


def bigtile_profile(enable: bool) -> None:
    """Start / stop the in-situ timing of the big-tile core's launches (bench.py's matrix-roofline leg)."""
    check(lib.stemgnn_profile_bigtile(1 if enable else 0), "profile_bigtile")


def bigtile_profile_collect() -> Tuple[float, float, int]:
    """-> (kernel ms, executed flop, launches) of the core's launches since the last collect."""
    ms, flop, n = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_int64(0)
    check(lib.stemgnn_profile_bigtile_collect(ctypes.byref(ms), ctypes.byref(flop), ctypes.byref(n)), "profile_bigtile_collect")
    return float(ms.value), float(flop.value), int(n.value)


def linear_release_scratch() -> None:
    """Unregister and free every arena (tests; a process that is done with the large configurations)."""
    for (dev, st) in list(_BT_ARENA):
        with torch.cuda.device(dev):
            check(lib.stemgnn_linear_set_scratch(None, 0, st), "linear_set_scratch")
        del _BT_ARENA[(dev, st)]


def linear_fwd(x1: Tensor, w1: Tensor, x2: Optional[Tensor], w2: Optional[Tensor], bias: Optional[Tensor],
               want_stats: bool = False, x1_rows: int = -1):
    """y = x1 w1^T (+ x2 w2^T) + bias; optionally the per-row-block column partials of y.  ``x1_rows`` >= 0
    promises that rows >= x1_rows of x1 are zero (row tiles past them skip x1's half of the contraction)."""
    _req(x1, torch.float32, "x1", 2)
    _req(w1, torch.float32, "w1", 2)
    M, K1 = x1.shape
    N = w1.size(0)
    if w1.size(1) != K1:
        raise RuntimeError(f"linear: x1 {tuple(x1.shape)} vs w1 {tuple(w1.shape)}")
    K2 = 0
    if x2 is not None:
        _req(x2, torch.float32, "x2", 2)
        _req(w2, torch.float32, "w2", 2)
        K2 = x2.size(1)
        if 0 <= x1_rows <= x1.size(0) and x1.size(0) < x2.size(0):
            M = x2.size(0)  # x1 holds its x1_rows meaningful rows only (rows past them count as zero, never read)
        if x2.size(0) != M or tuple(w2.shape) != (N, K2) or (x1.size(0) != M and x1.size(0) < x1_rows):
            raise RuntimeError("linear: second operand pair has inconsistent shapes")
    if bias is not None:
        _req(bias, torch.float32, "bias", 1)
    y = torch.empty(M, N, dtype=torch.float32, device=x1.device)
    partial = None
    blocks = int(lib.stemgnn_linear_stats_blocks(M, N))
    if want_stats:
        partial = torch.empty(max(blocks, 1), 2, N, dtype=torch.float32, device=x1.device)
    linear_scratch(M, K1 + K2, N)
    check(lib.stemgnn_linear_fwd(_p(x1), _p(w1), K1, _p(x2), _p(w2), K2, _p(bias), M, N, _p(y), _p(partial), None,
                                 int(x1_rows), _stream()), "linear_fwd")
    return y, partial, blocks


def linear_bwd_weight(dy: Tensor, x: Tensor, want_bias: bool):
    """dw = dy^T x, db = colsum(dy) (deterministic split reduction)."""
    M, N = dy.shape
    K = x.size(1)
    dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_bias else None
    ws = _workspace(lib.stemgnn_linear_bwd_weight_workspace_bytes(M, N, K), dy.device)
    linear_scratch(M, N, K)
    check(lib.stemgnn_linear_bwd_weight(_p(dy), _p(x), M, N, K, _p(dw), _p(db), _p(ws), ws.numel(), _stream()),
          "linear_bwd_weight")
    return dw, db


def linear_bwd_data(dy: Tensor, w: Tensor) -> Tensor:
    """dx = dy w for y = x w^T (w [N, K] as stored)."""
    _req(dy, torch.float32, "dy", 2)
    _req(w, torch.float32, "w", 2)
    M, N = dy.shape
    if w.size(0) != N:
        raise RuntimeError(f"linear_bwd_data: dy has {N} columns, w has {w.size(0)} rows")
    dx = torch.empty(M, w.size(1), dtype=torch.float32, device=dy.device)
    linear_scratch(M, N, w.size(1))
    check(lib.stemgnn_linear_bwd_data(_p(dy), _p(w), M, N, w.size(1), _p(dx), _stream()), "linear_bwd_data")
    return dx


def transpose(w: Tensor) -> Tensor:
    _req(w, torch.float32, "w", 2)
    out = torch.empty(w.size(1), w.size(0), dtype=torch.float32, device=w.device)
    check(lib.stemgnn_transpose(_p(w), w.size(0), w.size(1), _p(out), _stream()), "transpose")
    return out


class LinearFn(torch.autograd.Function):
    """y = x1 w1^T (+ x2 w2^T) + b on the matrix cores (fp32 result, csrc/linear.hip); returns (y, column partials or None).
    Replaces nn.Linear / lin_l + lin_r (reference model/encoder.py:83-87, model/vq.py:881,1041).

    ``x1_rows`` (>= 0): rows >= x1_rows of x1 are zero by construction (the aggregate of a sampled batch).  The forward
    skips x1's half of the contraction past them; the backward computes x1's gradient for rows < x1_rows only (the
    rest of the returned buffer is NOT written: its only consumer, the aggregation backward, never reads it) and
    contracts w1's gradient over those rows only."""

    @staticmethod
    def forward(ctx, x1, w1, x2, w2, bias, want_stats, x1_rows=-1):
        x1 = x1.contiguous()
        w1c = w1.contiguous()
        x2c = None if x2 is None else x2.contiguous()
        w2c = None if w2 is None else w2.contiguous()
        rows = int(x1_rows)
        if rows < 0 or rows >= x1.size(0) or x2c is None:
            rows = -1  # without a second operand every row still needs its bias: no saving, keep the plain path
        y, partial, blocks = linear_fwd(x1, w1c, x2c, w2c, bias, want_stats, rows)
        ctx.save_for_backward(x1, w1c, x2c, w2c)
        ctx.has_bias = bias is not None
        ctx.x1_rows = rows
        if partial is not None:
            ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    def backward(ctx, gy, _gpartial):
        x1, w1, x2, w2 = ctx.saved_tensors
        gy = gy.contiguous()
        need = ctx.needs_input_grad
        rows = ctx.x1_rows
        gx1 = gw1 = gx2 = gw2 = gb = None
        if rows >= 0:
            gy1, x1p = gy[:rows], x1[:rows]  # leading rows: contiguous views
            if need[0]:
                gx1 = torch.empty_like(x1)
                gx1[rows:].zero_()  # rows past x1_rows do not reach the output: their gradient is zero, not garbage
                if rows > 0:
                    linear_scratch(rows, gy.size(1), w1.size(1))
                    check(lib.stemgnn_linear_bwd_data(_p(gy1), _p(w1), rows, gy.size(1), w1.size(1), _p(gx1), _stream()),
                          "linear_bwd_data")
            if need[1]:
                if rows > 0:
                    gw1, _ = linear_bwd_weight(gy1, x1p, False)
                else:
                    gw1 = torch.zeros_like(w1)
        else:
            if need[0]:
                gx1 = linear_bwd_data(gy, w1)
            if need[1]:
                gw1, gb = linear_bwd_weight(gy, x1, ctx.has_bias and need[4])
        if x2 is not None:
            if need[2]:
                gx2 = linear_bwd_data(gy, w2)
            if need[3]:
                gw2, gb2 = linear_bwd_weight(gy, x2, ctx.has_bias and need[4] and gb is None)
                gb = gb if gb is not None else gb2
        if ctx.has_bias and need[4] and gb is None:
            gb = gy.sum(dim=0)
        return gx1, gw1, gx2, gw2, gb, None, None


class MatmulFn(torch.autograd.Function):
    """y [M, N] = a [M, K] @ w [K, N] with the weight's contraction index slow (the K-expert product of
    MixtureSageLayer, reference model/encoder.py:126: einsum('nd,kdo->nko') is this with w = the experts laid side by
    side).  Forward is the backward-data tile (weight read as stored), the two gradients are the forward and the
    weight-gradient tiles -- no transposed copies."""

    @staticmethod
    def forward(ctx, a, w):
        a, w = a.contiguous(), w.contiguous()
        ctx.save_for_backward(a, w)
        return linear_bwd_data(a, w)

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        g = g.contiguous()
        ga = gw = None
        if ctx.needs_input_grad[0]:
            ga = linear_fwd(g, w, None, None, None)[0]      # g w^T
        if ctx.needs_input_grad[1]:
            gw, _ = linear_bwd_weight(a, g, False)           # a^T g
        return ga, gw


def linear(x: Tensor, lin: "torch.nn.Linear") -> Tensor:
    """nn.Linear forward through LinearFn (any leading dims)."""
    lead = x.shape[:-1]
    y, _ = LinearFn.apply(x.reshape(-1, x.shape[-1]), lin.weight, None, None, lin.bias, False)
    return y.reshape(*lead, y.shape[-1])


# ----------------------------------------------------------------------------------------
# K6-K8, K10
# ----------------------------------------------------------------------------------------
class VqAssignFn(torch.autograd.Function):
    """(quant, ind, mse) = cosine-codebook assignment of xp [N, H*Dc] against embed [H, K, Dc]
    (reference model/vq.py:891, 650-657, 931-937, 1007).  mse = loss_weight * mean((q - xn)^2) carries the
    commitment gradient; embed receives no gradient through this op (q is detached in the
    reference: vq.py:931-937 with VectorQuantize.learnable_codebook == False)."""

    @staticmethod
    def forward(ctx, xp, embed, heads, training, snapshot_embed=False, loss_weight=1.0):
        xp = xp.contiguous()
        embed_c = embed.detach().contiguous()
        if snapshot_embed:
            # the EMA update rewrites `embed` in place right after this op (vq.py:682) while the
            # backward still needs the codes that were actually assigned
            embed_c = embed_c.clone()
        _req(xp, torch.float32, "xp", 2)
        _req(embed_c, torch.float32, "embed", 3)
        N = xp.size(0)
        H, K, Dc = embed_c.shape
        if H != heads or xp.size(1) != H * Dc:
            raise RuntimeError(f"xp shape {tuple(xp.shape)} does not match codebook {tuple(embed_c.shape)}")
        dev = xp.device
        norm = torch.empty(N, H, dtype=torch.float32, device=dev)
        ind = torch.empty(N, H, dtype=torch.int64, device=dev)
        quant = torch.empty_like(xp)
        sqerr = torch.empty(1, dtype=torch.float32, device=dev)
        ws = _workspace(lib.stemgnn_vq_workspace_bytes(N, H, Dc, K), dev)
        numel = max(N * H * Dc, 1)
        linear_scratch(N, Dc, Dc, vq=(H, Dc, K))
        check(lib.stemgnn_vq_assign_fwd(_p(xp), N, H, Dc, _p(embed_c), K, int(bool(training)), None, _p(norm), _p(ind),
                                        _p(quant), _p(sqerr), float(loss_weight) / float(numel), _p(ws), ws.numel(),
                                        _stream()), "vq_assign_fwd")
        ctx.save_for_backward(xp, norm, ind, embed_c)
        ctx.mark_non_differentiable(ind)
        ctx.loss_weight = float(loss_weight)
        return quant, ind, sqerr

    @staticmethod
    def backward(ctx, g_quant, _g_ind, g_mse):
        xp, norm, ind, embed = ctx.saved_tensors
        N = xp.size(0)
        H, K, Dc = embed.shape
        if g_quant is None:
            g_quant = torch.zeros_like(xp)
        g_quant = g_quant.contiguous()
        g_loss = None if g_mse is None else g_mse.contiguous().float()
        g_xp = torch.empty_like(xp)
        check(lib.stemgnn_vq_assign_bwd(_p(g_quant), _p(g_loss), ctx.loss_weight, _p(xp), _p(norm), _p(ind), _p(embed), N,
                                        H, Dc, K, _p(g_xp), _stream()), "vq_assign_bwd")
        return g_xp, None, None, None, None, None


class VqFn(torch.autograd.Function):
    """VectorQuantize.forward (reference model/vq.py:849-1064) for callers that use (quantize, embed_ind, loss) and
    not the per-head codes, as ONE library call per direction (csrc/phases.hip: stemgnn_vq_fwd / _bwd): project_in,
    fused cosine assignment, commitment + orthogonal terms, and project_out read off the table of projected code
    rows; in the backward project_out's weight gradient comes from per-code segment sums.  The codebook receives the
    orthogonal term's gradient only (the quantised rows are detached in the reference, vq.py:931-937)."""

    @staticmethod
    def _params(z, embed, w_in, b_in, w_out, b_out, cfg, grads=None):
        H, K, Dc = embed.shape
        p = VqParams()
        p.dim, p.heads, p.code_dim, p.codebook_size = z.size(1), H, Dc, K
        p.w_in, p.b_in, p.w_out, p.b_out, p.embed = _p(w_in), _p(b_in), _p(w_out), _p(b_out), _p(embed)
        p.commitment_weight, p.ortho_weight = float(cfg["commit"]), float(cfg["ortho"])
        ids = cfg["ortho_ids"]
        p.ortho_ids, p.num_ortho_ids = _p(ids), 0 if ids is None else ids.numel()
        if grads is not None:
            p.g_w_in, p.g_b_in, p.g_w_out, p.g_b_out, p.g_embed = (_p(t) for t in grads)
        return p

    @staticmethod
    def forward(ctx, z, embed, w_in, b_in, w_out, b_out, cfg):
        z = z.contiguous()
        tensors = [embed, w_in, b_in, w_out, b_out]
        for t in (z, *tensors):
            if t is not None:
                _req(t.detach(), torch.float32, "vq operand")
        if embed.dim() != 3 or tuple(w_in.shape) != (embed.size(0) * embed.size(2), z.size(1)) or \
                tuple(w_out.shape) != (z.size(1), embed.size(0) * embed.size(2)):
            raise RuntimeError("vq phase: projection / codebook shapes do not match")
        if cfg["ortho_ids"] is not None:
            _req(cfg["ortho_ids"], torch.int64, "ortho_ids", 1)
        p = VqFn._params(z, embed, w_in, b_in, w_out, b_out, cfg)
        N, H = z.size(0), embed.size(0)
        save = _workspace(lib.stemgnn_vq_save_bytes(ctypes.byref(p), N), z.device)
        quantize = torch.empty_like(z)
        ind = torch.empty(N, H, dtype=torch.int64, device=z.device)
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        linear_scratch(N, z.size(1), w_in.size(0), vq=(H, embed.size(2), embed.size(1)))
        check(lib.stemgnn_vq_fwd(ctypes.byref(p), _p(z), N, int(bool(cfg["training"])), _p(quantize), _p(ind), _p(loss),
                                 _p(save), save.numel(), _stream()), "vq_fwd")
        ctx.cfg = cfg
        ctx.save_for_backward(z, ind, save, embed, w_in, b_in, w_out, b_out)
        ctx.mark_non_differentiable(ind)
        ctx.set_materialize_grads(False)  # no zeros_like(ind) launch per backward; None gradients are handled below
        return quantize, ind, loss

    @staticmethod
    def backward(ctx, g_quantize, _g_ind, g_loss):
        z, ind, save, embed, w_in, b_in, w_out, b_out = ctx.saved_tensors
        need = ctx.needs_input_grad
        tensors = [w_in, b_in, w_out, b_out, embed]
        wanted = [need[2], need[3], need[4], need[5], need[1]]
        if wanted[1] and not wanted[0]:
            wanted[0] = True  # project_in's bias gradient is produced with its weight gradient
        grads = [torch.empty_like(t) if (t is not None and w) else None for t, w in zip(tensors, wanted)]
        p = VqFn._params(z, embed, w_in, b_in, w_out, b_out, ctx.cfg, grads)
        N = z.size(0)
        g_z = torch.empty_like(z) if need[0] else None
        scratch = _workspace(lib.stemgnn_vq_bwd_scratch_bytes(ctypes.byref(p), N), z.device)
        gq = None if g_quantize is None else g_quantize.contiguous()
        gl = None if g_loss is None else g_loss.reshape(1).contiguous().float()
        linear_scratch(N, z.size(1), w_in.size(0), factor=2)
        check(lib.stemgnn_vq_bwd(ctypes.byref(p), _p(z), N, _p(ind), _p(gq), _p(gl), _p(g_z), _p(save), save.numel(),
                                 _p(scratch), scratch.numel(), _stream()), "vq_bwd")
        g_w_in, g_b_in, g_w_out, g_b_out, g_embed = grads
        return (g_z, g_embed if need[1] else None, g_w_in if need[2] else None, g_b_in if need[3] else None,
                g_w_out if need[4] else None, g_b_out if need[5] else None, None)


def vq_ema_stats(xp: Tensor, norm: Tensor, ind: Tensor, codebook_size: int) -> Tuple[Tensor, Tensor]:
    """bins [H, K], embed_sum [H, K, Dc] of the normalised rows per assigned code (vq.py:661-672)."""
    _req(xp, torch.float32, "xp", 2)
    _req(norm, torch.float32, "norm", 2)
    _req(ind, torch.int64, "ind", 2)
    N, H = norm.shape
    Dc = xp.size(1) // H
    K = int(codebook_size)
    dev = xp.device
    bins = torch.empty(H, K, dtype=torch.float32, device=dev)
    embed_sum = torch.empty(H, K, Dc, dtype=torch.float32, device=dev)
    ws = _workspace(lib.stemgnn_vq_ema_workspace_bytes(N, H, Dc, K), dev)
    check(lib.stemgnn_vq_ema_stats(_p(xp), _p(norm), _p(ind), N, H, Dc, K, _p(bins), _p(embed_sum), _p(ws), ws.numel(),
                                   _stream()), "vq_ema_stats")
    return bins, embed_sum


def vq_norms(xp: Tensor, heads: int) -> Tensor:
    n = xp.size(0)
    return xp.view(n, heads, -1).norm(dim=-1)


# ----------------------------------------------------------------------------------------
# K11 / K12 / lookups / K14
# ----------------------------------------------------------------------------------------
class EdgeDotFn(torch.autograd.Function):
    """out[e] = <z[u_e], z[v_e]> (reference model/encoder.py:365)."""

    @staticmethod
    def forward(ctx, z, edge_index):
        z = z.contiguous()
        _req(z, torch.float32, "z", 2)
        ei = _req(edge_index.contiguous(), torch.int64, "edge_index", 2)
        E = ei.size(1)
        out = torch.empty(E, dtype=torch.float32, device=z.device)
        check(lib.stemgnn_edge_dot_fwd(_p(z), z.size(0), z.size(1), _p(ei), E, _p(out), _stream()), "edge_dot_fwd")
        ctx.save_for_backward(z, ei)
        return out

    @staticmethod
    def backward(ctx, g_out):
        z, ei = ctx.saved_tensors
        g_z = torch.zeros_like(z)
        check(lib.stemgnn_edge_dot_bwd(_p(g_out.contiguous()), _p(z), z.size(0), z.size(1), _p(ei), ei.size(1),
                                       _p(g_z), _stream()), "edge_dot_bwd")
        return g_z, None


class EdgeBceLossFn(torch.autograd.Function):
    """topo_recon_loss (reference model/pt_model.py:62-65): mean -log(sigmoid(<z_u,z_v>)+EPS) over the
    first `num_pos` edges + mean -log(1-sigmoid(.)+EPS) over the rest, as three small kernels
    forward (edge scores, loss + d loss/d score, -) and one scaled scatter backward."""

    @staticmethod
    def forward(ctx, z, edge_index, num_pos):
        z = z.contiguous()
        _req(z, torch.float32, "z", 2)
        ei = _req(edge_index.contiguous(), torch.int64, "edge_index", 2)
        E = ei.size(1)
        dots = torch.empty(E, dtype=torch.float32, device=z.device)
        check(lib.stemgnn_edge_dot_fwd(_p(z), z.size(0), z.size(1), _p(ei), E, _p(dots), _stream()), "edge_dot_fwd")
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        coef = torch.empty_like(dots)
        check(lib.stemgnn_edge_bce_loss(_p(dots), int(num_pos), E - int(num_pos), _p(loss), _p(coef), _stream()),
              "edge_bce_loss")
        ctx.save_for_backward(z, ei, coef)
        return loss.view(())

    @staticmethod
    def backward(ctx, g_loss):
        z, ei, coef = ctx.saved_tensors
        g_z = torch.zeros_like(z)
        g = g_loss.reshape(1).contiguous().float()
        check(lib.stemgnn_edge_dot_bwd_scaled(_p(coef), _p(g), _p(z), z.size(0), z.size(1), _p(ei), ei.size(1), _p(g_z),
                                              _stream()), "edge_dot_bwd_scaled")
        return g_z, None, None


class EdgeConcatFn(torch.autograd.Function):
    """out[e] = cat(z[u_e], z[v_e]) (reference model/pt_model.py:80)."""

    @staticmethod
    def forward(ctx, z, edge_index):
        z = z.contiguous()
        _req(z, torch.float32, "z", 2)
        ei = _req(edge_index.contiguous(), torch.int64, "edge_index", 2)
        E, D = ei.size(1), z.size(1)
        out = torch.empty(E, 2 * D, dtype=torch.float32, device=z.device)
        check(lib.stemgnn_edge_concat_fwd(_p(z), z.size(0), D, _p(ei), E, _p(out), _stream()), "edge_concat_fwd")
        ctx.save_for_backward(ei)
        ctx.shape = (z.size(0), D)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (ei,) = ctx.saved_tensors
        N, D = ctx.shape
        g_z = torch.zeros(N, D, dtype=torch.float32, device=g_out.device)
        check(lib.stemgnn_edge_concat_bwd(_p(g_out.contiguous()), N, D, _p(ei), ei.size(1), _p(g_z), _stream()),
              "edge_concat_bwd")
        return g_z, None


class MseLossFn(torch.autograd.Function):
    """F.mse_loss(pred, target) (reference model/pt_model.py:43,81): one kernel forward, one backward."""

    @staticmethod
    def forward(ctx, pred, target):
        pred = pred.contiguous()
        target = target.contiguous()
        _req(pred, torch.float32, "pred")
        _req(target, torch.float32, "target")
        if pred.shape != target.shape:
            raise RuntimeError(f"mse_loss: shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        ws = _workspace(lib.stemgnn_loss_workspace_bytes(256), pred.device)
        check(lib.stemgnn_mse_loss_fwd(_p(pred), _p(target), pred.numel(), 1.0, _p(loss), _p(ws), ws.numel(), _stream()),
              "mse_loss_fwd")
        ctx.save_for_backward(pred, target)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        gp = torch.empty_like(pred)
        check(lib.stemgnn_mse_loss_bwd(_p(pred), _p(target), pred.numel(), 1.0, _p(g.reshape(1).contiguous()), _p(gp),
                                       _stream()), "mse_loss_bwd")
        return gp, None


class CosineLossFn(torch.autograd.Function):
    """mean(1 - cos(z, h)) with z detached (reference model/pt_model.py:93-100)."""

    @staticmethod
    def forward(ctx, z, h):
        z = z.contiguous()
        h = h.contiguous()
        _req(z, torch.float32, "z", 2)
        _req(h, torch.float32, "h", 2)
        rows, D = h.shape
        loss = torch.empty(1, dtype=torch.float32, device=h.device)
        save = torch.empty(max(rows, 1), 3, dtype=torch.float32, device=h.device)
        ws = _workspace(lib.stemgnn_loss_workspace_bytes(rows), h.device)
        check(lib.stemgnn_cosine_loss_fwd(_p(z), _p(h), rows, D, 1.0, _p(loss), _p(save), _p(ws), ws.numel(), _stream()),
              "cosine_loss_fwd")
        ctx.save_for_backward(z, h, save)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        z, h, save = ctx.saved_tensors
        gh = torch.empty_like(h)
        check(lib.stemgnn_cosine_loss_bwd(_p(z), _p(h), h.size(0), h.size(1), 1.0, _p(g.reshape(1).contiguous()),
                                          _p(save), _p(gh), _stream()), "cosine_loss_bwd")
        return None, gh


class OrthoLossFn(torch.autograd.Function):
    """orthogonal_loss_fn(embed[:, ids]) * weight (reference model/vq.py:232-237,1011-1028)."""

    @staticmethod
    def forward(ctx, embed, ids, weight):
        e = embed.contiguous()
        _req(e, torch.float32, "embed", 3)
        ids = _req(ids.contiguous(), torch.int64, "ids", 1)
        H, K, Dc = e.shape
        loss = torch.empty(1, dtype=torch.float32, device=e.device)
        ws = _workspace(lib.stemgnn_loss_workspace_bytes(H * ids.numel()), e.device)
        check(lib.stemgnn_ortho_loss_fwd(_p(e), _p(ids), H, K, Dc, ids.numel(), float(weight), _p(loss), _p(ws),
                                         ws.numel(), _stream()), "ortho_loss_fwd")
        ctx.save_for_backward(e, ids)
        ctx.weight = float(weight)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        e, ids = ctx.saved_tensors
        H, K, Dc = e.shape
        ge = torch.empty_like(e)
        check(lib.stemgnn_ortho_loss_bwd(_p(e), _p(ids), H, K, Dc, ids.numel(), ctx.weight,
                                         _p(g.reshape(1).contiguous()), _p(ge), _stream()), "ortho_loss_bwd")
        return ge, None, None


class QueryFanOutFn(torch.autograd.Function):
    """The three consumers of the decoder query in PretrainModel.forward (reference
    model/pt_model.py:128-131): q itself (topology decoder), its first `bs` rows (feature and
    semantic terms) and cat([q[u], q[v]]) over the sampled edges (topo-sem term).  Forward is a
    slice copy + the edge-concat kernel.  Backward folds the three gradients into ONE dense
    buffer: the sparse contributions are scattered / added into the dense one in place, instead
    of autograd materialising three zero-filled [N, D] tensors and summing them with three
    full-size add kernels."""

    @staticmethod
    def forward(ctx, q, bs, edge_index):
        q = q.contiguous()
        _req(q, torch.float32, "q", 2)
        ei = _req(edge_index.contiguous(), torch.int64, "edge_index", 2)
        N, D = q.shape
        E = ei.size(1)
        head = q[:bs].clone()
        zz = torch.empty(E, 2 * D, dtype=torch.float32, device=q.device)
        check(lib.stemgnn_edge_concat_fwd(_p(q), N, D, _p(ei), E, _p(zz), _stream()), "edge_concat_fwd")
        ctx.save_for_backward(ei)
        ctx.meta = (N, D, int(head.size(0)))
        return q.view_as(q), head, zz

    @staticmethod
    def backward(ctx, g_q, g_head, g_zz):
        (ei,) = ctx.saved_tensors
        N, D, nh = ctx.meta
        if g_q is None:
            g = torch.zeros(N, D, dtype=torch.float32, device=ei.device)
        else:
            # an incoming gradient is not this node's to mutate (another consumer, a hook or retain_grad may hold it):
            # the scatter below accumulates into a copy
            g = g_q.clone(memory_format=torch.contiguous_format)
        if g_zz is not None:
            check(lib.stemgnn_edge_concat_bwd(_p(g_zz.contiguous()), N, D, _p(ei), ei.size(1), _p(g), _stream()),
                  "edge_concat_bwd")
        if g_head is not None:
            g[:nh] += g_head
        return g, None, None


class HeadsFn(torch.autograd.Function):
    """The four reconstruction heads that read the decoder query in PretrainModel.forward (reference
    model/pt_model.py:39-102,128-131) as ONE library call per direction (csrc/heads.hip): edge sampling, negative
    sampling, the decoders' products and the four losses; backward: every head's gradient folded into one dense
    [N, D] buffer plus the decoder parameter gradients.  Returns (losses [4] = feat, topo, topo_sem, sem -- unweighted,
    draws dict).  ``params`` = (w_feat, b_feat, w_topo, b_topo, w_ts, b_ts, w_sem, b_sem)."""

    @staticmethod
    def _struct(q, x_feat, params, grads=None):
        p = HeadsParams()
        p.dim, p.in_dim = q.size(1), x_feat.size(1)
        (p.w_feat, p.b_feat, p.w_topo, p.b_topo, p.w_ts, p.b_ts, p.w_sem, p.b_sem) = (_p(t) for t in params)
        if grads is not None:
            (p.g_w_feat, p.g_b_feat, p.g_w_topo, p.g_b_topo, p.g_w_ts, p.g_b_ts, p.g_w_sem, p.g_b_sem) = \
                (_p(t) for t in grads)
        return p

    @staticmethod
    def forward(ctx, q, graph, etab, etype, x_feat, z_teacher, bs, k, keys, out, *params):
        q = q.contiguous()
        _req(q, torch.float32, "query", 2)
        _req(x_feat, torch.float32, "x", 2)
        _req(z_teacher, torch.float32, "teacher", 2)
        _req(etab, torch.float32, "edge_type_table", 2)
        _req(etype, torch.int64, "edge_type", 1)
        ei = _req(graph.edge_index, torch.int64, "edge_index", 2)
        N, D = q.shape
        E = ei.size(1)
        if x_feat.size(0) < bs or z_teacher.size(0) < bs or z_teacher.size(1) != D or etab.size(1) != D:
            raise RuntimeError("heads phase: operand shapes do not match the query")
        W = (params[0], params[2], params[4], params[6])
        want = ((x_feat.size(1), D), (D, D), (D, 2 * D), (D, D))
        if any(tuple(w.shape) != s for w, s in zip(W, want)):
            raise RuntimeError("heads phase: decoder shapes do not match (feat [in, D], topo [D, D], topo_sem [D, 2D], sem [D, D])")
        p = HeadsFn._struct(q, x_feat, params)
        gv = graph_view(graph, False)
        dev = q.device
        i64 = dict(dtype=torch.int64, device=dev)
        topo_perm, topo_edges = torch.empty(k, **i64), torch.empty(2, 2 * k, **i64)
        ts_perm, ts_edges, ts_type = torch.empty(k, **i64), torch.empty(2, k, **i64), torch.empty(k, **i64)
        losses = torch.empty(4, dtype=torch.float32, device=dev)
        save = _workspace(lib.stemgnn_heads_save_bytes(ctypes.byref(p), N, E, bs, k), dev)
        seed, (o1, o2, o3) = keys
        linear_scratch(max(N, k), 2 * q.size(1), q.size(1))
        check(lib.stemgnn_heads_fwd(ctypes.byref(p), ctypes.byref(gv), _p(ei), _p(etype), E, _p(etab), etab.size(0), _p(q),
                                    _p(x_feat), _p(z_teacher), bs, k, seed, o1, o2, o3, _p(topo_perm), _p(topo_edges),
                                    _p(ts_perm), _p(ts_edges), _p(ts_type), _p(losses), _p(save), save.numel(), _stream()),
              "heads_fwd")
        out["topo_perm"], out["neg_edge_index"], out["topo_sem_perm"] = topo_perm, topo_edges[:, k:], ts_perm
        ctx.meta = (bs, k, E)
        ctx.save_for_backward(q, x_feat, z_teacher, topo_edges, ts_edges, save, *params)
        # four outputs of this node (not select views of one output: their backward would be four zero fills, four
        # copies and three adds)
        return losses[0], losses[1], losses[2], losses[3]

    @staticmethod
    def backward(ctx, *g4):
        q, x_feat, z_teacher, topo_edges, ts_edges, save, *params = ctx.saved_tensors
        dev = q.device
        g4 = [torch.zeros((), device=dev) if g is None else g for g in g4]
        base = g4[0].data_ptr()
        if all(g.dtype == torch.float32 and g.is_cuda and g.data_ptr() == base + 4 * i for i, g in enumerate(g4)):
            g_losses = g4[0]   # consecutive elements of one buffer (what the weighted total's backward hands out)
        else:
            g_losses = torch.stack([g.reshape(()).float() for g in g4])
        bs, k, E = ctx.meta
        grads = [None if t is None else torch.empty_like(t) for t in params]
        p = HeadsFn._struct(q, x_feat, params, grads)
        N = q.size(0)
        g_q = torch.empty_like(q)
        scratch = _workspace(lib.stemgnn_heads_bwd_scratch_bytes(ctypes.byref(p), N, bs, k), q.device)
        linear_scratch(max(N, k), 2 * q.size(1), q.size(1), factor=3)
        check(lib.stemgnn_heads_bwd(ctypes.byref(p), N, _p(q), _p(x_feat), _p(z_teacher), bs, k, _p(topo_edges),
                                    _p(ts_edges), _p(g_losses), _p(g_q), _p(save), save.numel(), E, _p(scratch),
                                    scratch.numel(), _stream()), "heads_bwd")
        need = ctx.needs_input_grad
        return (g_q if need[0] else None, None, None, None, None, None, None, None, None, None,
                *[g if need[10 + i] else None for i, g in enumerate(grads)])


_CLIP_MAX = int(lib.stemgnn_clip_grad_max_tensors())


def clip_grad_norm_(parameters, max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) (L2; reference pretrain.py:62) in three launches.
    Returns the total norm (0-dim tensor).  Gradients that are not dense fp32 on one device, or more than the
    table holds, go through torch's own implementation."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.zeros(())
    dev = grads[0].device
    if (len(grads) > _CLIP_MAX or not grads[0].is_cuda
            or any(g.dtype != torch.float32 or g.device != dev or not g.is_contiguous() for g in grads)):
        return torch.nn.utils.clip_grad_norm_([p for p in parameters if p.grad is not None], max_norm)
    n = len(grads)
    ptrs = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
    sizes = (ctypes.c_int64 * n)(*[g.numel() for g in grads])
    total = sum(sizes)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    ws = _workspace(lib.stemgnn_clip_grad_workspace_bytes(total, n), dev)
    check(lib.stemgnn_clip_grad_norm(ptrs, sizes, n, float(max_norm), _p(out), _p(ws), ws.numel(), _stream()),
          "clip_grad_norm")
    return out[0]


class WeightedSumFn(torch.autograd.Function):
    """sum_i w_i * term_i of scalar device tensors (the lambda-weighted total loss, reference pretrain.py:51-58):
    one launch forward, one backward, instead of stack + mul + sum and their three backward kernels."""

    @staticmethod
    def forward(ctx, weights, *terms):
        n = len(terms)
        ts = [t.reshape(1).contiguous() for t in terms]
        for t in ts:  # scalars: any 4-byte aligned device float will do (e.g. one element of a [4] loss vector)
            if not (t.is_cuda and t.dtype == torch.float32):
                raise RuntimeError("loss term: expected a CUDA float32 scalar")
        out = torch.empty(1, dtype=torch.float32, device=ts[0].device)
        ctx.w = (ctypes.c_float * n)(*[float(w) for w in weights])
        ctx.n = n
        ctx.shapes = [t.shape for t in terms]
        check(lib.stemgnn_weighted_sum((ctypes.c_void_p * n)(*[t.data_ptr() for t in ts]), ctx.w, n, _p(out), _stream()),
              "weighted_sum")
        return out

    @staticmethod
    def backward(ctx, g):
        if g.data_ptr() in UNIT_GRADIENTS:
            # the seed of loss.backward() is a registered constant 1.0 (pretrain._ones_like_loss): the terms' gradients
            # are the weights themselves -- a cached device tensor, no launch
            key = (g.device, tuple(ctx.w))
            gt = _WEIGHT_GRADS.get(key)
            if gt is None:
                gt = _WEIGHT_GRADS[key] = torch.tensor(list(ctx.w), dtype=torch.float32, device=g.device)
        else:
            gt = torch.empty(ctx.n, dtype=torch.float32, device=g.device)
            check(lib.stemgnn_weighted_sum_bwd(ctx.w, ctx.n, _p(g.reshape(1).contiguous()), _p(gt), _stream()),
                  "weighted_sum_bwd")
        return (None, *[gt[i].reshape(shape) for i, shape in enumerate(ctx.shapes)])


# data pointers of device tensors that hold the constant 1.0 and are only ever used as the seed gradient of a backward
# pass (registered by their owner, never written again); weight vectors of WeightedSumFn as device tensors
UNIT_GRADIENTS = set()
_WEIGHT_GRADS = {}


def _grad_table(grads):
    n = len(grads)
    return ((ctypes.c_void_p * n)(*[g.data_ptr() for g in grads]), (ctypes.c_int64 * n)(*[g.numel() for g in grads]), n)


def grad_norm_coef(grads, max_norm: float) -> Tensor:
    """-> device tensor [total L2 norm, min(1, max_norm / (norm + 1e-6))] of dense fp32 gradients on one device
    (at most `_CLIP_MAX` tensors); nothing is scaled."""
    dev = grads[0].device
    ptrs, sizes, n = _grad_table(grads)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    ws = _workspace(lib.stemgnn_clip_grad_workspace_bytes(sum(sizes), n), dev)
    check(lib.stemgnn_grad_norm_coef(ptrs, sizes, n, float(max_norm), _p(out), _p(ws), ws.numel(), _stream()),
          "grad_norm_coef")
    return out


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW (amsgrad off, maximize off; reference pretrain.py:134-136) as ONE launch per parameter
    group over a table of the tensors, with an optional device-side gradient factor (the clipping coefficient of
    ``grad_norm_coef``) applied while the gradients are read.  Same update arithmetic as ATen's fused kernel;
    ``state[p]`` holds ``step`` (python int), ``exp_avg``, ``exp_avg_sq``."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_coef: Optional[Tensor] = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if (p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous()
                        or p.grad.dtype != torch.float32):
                    raise RuntimeError("FusedAdamW: dense contiguous fp32 CUDA parameters and gradients only")
            b1, b2 = group["betas"]
            for lo in range(0, len(ps), _CLIP_MAX):
                chunk = ps[lo:lo + _CLIP_MAX]
                n = len(chunk)
                steps = {self.state[p]["step"] for p in chunk}
                if len(steps) != 1:
                    raise RuntimeError("FusedAdamW: parameters of one group must share their step count")
                step = steps.pop() + 1
                arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
                check(lib.stemgnn_adamw_step(arr(chunk), arr([p.grad for p in chunk]),
                                             arr([self.state[p]["exp_avg"] for p in chunk]),
                                             arr([self.state[p]["exp_avg_sq"] for p in chunk]),
                                             (ctypes.c_int64 * n)(*[p.numel() for p in chunk]), n, float(group["lr"]),
                                             float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                             step, _p(grad_coef), _stream()), "adamw_step")
                for p in chunk:
                    self.state[p]["step"] = step
        return loss


_validate_gather = True


def set_gather_validation(flag: bool) -> None:
    """Check every ``gather_rows`` index against the table (one device->host read per call) and raise IndexError
    like the reference's ``node_text_feat[data.x]`` does.  ``graph.set_validation`` switches this together with the
    edge_index range check; a loader that produces its own indices turns both off."""
    global _validate_gather
    _validate_gather = bool(flag)


def gather_rows(table: Tensor, index: Tensor, validate: Optional[bool] = None, capacity: int = 0) -> Tensor:
    """out[i] = table[index[i]] (device-side node_text_feat[x] of reference pretrain.py:33-38), in the table's dtype
    (fp32, or bf16 when the features are stored as bf16).  Out-of-range indices raise IndexError when validation is on
    (the default); with validation off their rows read as zeros.  ``capacity`` (rows): the result is the leading part
    of an allocation of that many rows -- a loader whose batches differ slightly in size then asks the caching
    allocator for the same block size every time (a request larger than any cached block goes to hipMalloc: ~8 ms)."""
    _req(table, table.dtype if table.dtype == torch.bfloat16 else torch.float32, "table", 2)
    _req(index, torch.int64, "index", 1)
    out = torch.empty(max(index.numel(), int(capacity)), table.size(1), dtype=table.dtype,
                      device=table.device)[:index.numel()]
    check_range = _validate_gather if validate is None else validate
    bad = torch.empty(1, dtype=torch.int32, device=table.device) if check_range else None
    check(lib.stemgnn_gather_rows_k(_p(table), _kind(table), table.size(0), table.size(1), _p(index), index.numel(),
                                    _p(out), _p(bad), _stream()), "gather_rows")
    if bad is not None:
        n_bad = int(bad.item())
        if n_bad:
            raise IndexError(f"gather_rows: {n_bad} of {index.numel()} indices lie outside [0, {table.size(0)})")
    return out


def ema_lerp_(teacher_flat: Tensor, student_flat: Tensor, decay: float) -> None:
    """teacher = teacher * decay + student * (1 - decay), in place (reference pt_model.py:104-106)."""
    _req(teacher_flat, torch.float32, "teacher", 1)
    _req(student_flat, torch.float32, "student", 1)
    if teacher_flat.numel() != student_flat.numel():
        raise RuntimeError("ema_lerp_: size mismatch")
    check(lib.stemgnn_ema_lerp(_p(teacher_flat), _p(student_flat), teacher_flat.numel(), float(decay), _stream()),
          "ema_lerp")
