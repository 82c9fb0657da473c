"""Generate golden vectors for the vector quantiser by IMPORTING the reference's
``STEM-GNN/model/vq.py`` (pure torch + einops).  Run in the build container only:

    python tests/golden/gen_vq_golden.py

The reference never travels to the GPU box; only the small ``vq_*.pt`` fixtures
written next to this script are committed.  Each fixture is a flat dict of tensors
(loadable with ``torch.load(..., weights_only=True)``): constructor arguments, the
initial state_dict, the input, the orthogonal-loss code ids the reference drew,
and the reference's outputs / gradients / post-step EMA buffers.
"""
import os
import sys
import warnings

import torch

warnings.filterwarnings("ignore")
sys.path.insert(0, "/root/reference/STEM-GNN")
from model.vq import VectorQuantize  # noqa: E402  (reference import: build container only)

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (N, D, H, K, Dc, ortho_max)
    "small": (131, 32, 4, 16, 32, 8),
    "mid": (160, 48, 4, 128, 32, 32),
    "onehead": (97, 48, 1, 32, 48, 32),   # heads=1, codebook_dim == dim -> no projections, K == ortho_max
}


def build(D, H, K, Dc, ortho_max, ema_update):
    return VectorQuantize(
        dim=D, codebook_size=K, codebook_dim=Dc, heads=H, separate_codebook_per_head=True,
        decay=0.8, commitment_weight=10, use_cosine_sim=True, orthogonal_reg_weight=1,
        orthogonal_reg_max_codes=ortho_max, orthogonal_reg_active_codes_only=False,
        kmeans_init=False, ema_update=ema_update)  # pretrain.py:104-119 argument pattern


def main():
    for name, (N, D, H, K, Dc, ortho_max) in CASES.items():
        for seed in (0, 1, 2):
            for ema in (False, True):
                torch.manual_seed(1000 + seed)
                vq = build(D, H, K, Dc, ortho_max, ema)
                state0 = {k: v.clone() for k, v in vq.state_dict().items()}
                z = torch.randn(N, D) * 1.5
                fx = {"meta": torch.tensor([N, D, H, K, Dc, ortho_max, int(ema), seed])}
                fx.update({"state0." + k: v for k, v in state0.items()})
                fx["z"] = z.clone()
                # ---- train-mode forward/backward
                vq.train()
                zt = z.clone().requires_grad_(True)
                torch.manual_seed(77 + seed)
                q, ind, loss, oq = vq(zt)
                torch.manual_seed(77 + seed)
                fx["ortho_ids"] = torch.randperm(K)[:ortho_max] if K > ortho_max else torch.arange(K)
                w = torch.linspace(-1.0, 1.0, q.numel()).view_as(q)  # fixed upstream gradient
                (loss.sum() + (q * w).sum()).backward()
                fx["train.quantize"], fx["train.embed_ind"] = q.detach().clone(), ind.clone()
                fx["train.loss"], fx["train.orig_quantize"] = loss.detach().clone(), oq.detach().clone()
                fx["train.grad_z"] = zt.grad.clone()
                for pn, p in vq.named_parameters():
                    if p.grad is not None:
                        fx["train.grad." + pn] = p.grad.clone()
                # top-2 similarity gap per (h, n), for the tie-aware index comparator
                with torch.no_grad():
                    x = vq.project_in(z).view(N, H, Dc).permute(1, 0, 2)
                    x = torch.nn.functional.normalize(x, dim=-1)
                    sim = torch.einsum('hnd,hcd->hnc', x, state0["_codebook.embed"])
                    top2 = sim.topk(2, dim=-1).values
                    fx["top2_gap"] = (top2[..., 0] - top2[..., 1]).permute(1, 0).contiguous()
                if ema:
                    for k, v in vq.state_dict().items():
                        if k.startswith("_codebook."):
                            fx["post." + k] = v.clone()
                # ---- eval-mode forward (on the initial state)
                vq2 = build(D, H, K, Dc, ortho_max, ema)
                vq2.load_state_dict(state0)
                vq2.eval()
                with torch.no_grad():
                    q2, ind2, loss2, oq2 = vq2(z)
                fx["eval.quantize"], fx["eval.embed_ind"] = q2.clone(), ind2.clone()
                fx["eval.loss"], fx["eval.orig_quantize"] = loss2.clone(), oq2.clone()
                path = os.path.join(HERE, f"vq_{name}_s{seed}_ema{int(ema)}.pt")
                torch.save(fx, path)
                print(path, os.path.getsize(path) // 1024, "KiB", "loss", float(loss))


if __name__ == "__main__":
    main()
