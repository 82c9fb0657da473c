"""Multi-dataset mix (BASELINE config 5 plumbing): union of member graphs and the weighted seed list, against the
oracle's restatement of reference dataset/process_datasets.py:166-198 (CPU; the product functions are torch ops)."""
import torch

from oracle import stem_oracle as O  # checker only


def test_merge_graphs_shifts_ids_like_the_reference():
    from stem_gnn_amd.data.multi import merge_graphs
    from stem_gnn_amd.data.synthetic import make_graph
    gs = [make_graph(50, 200, 8, 3, feat_rows=0, graph_seed=1), make_graph(70, 300, 8, 5, feat_rows=11, graph_seed=2),
          make_graph(20, 40, 8, 1, graph_seed=3)]
    u = merge_graphs(gs, ["a", "b", "c"])
    x, xe, ei, ptr = O.merge_member_graphs([dict(x=g.x, xe=g.xe, edge_index=g.edge_index, node_text_feat=g.node_text_feat,
                                                 edge_text_feat=g.edge_text_feat) for g in gs])
    assert torch.equal(u.x, x) and torch.equal(u.xe, xe) and torch.equal(u.edge_index, ei) and torch.equal(u.ptr, ptr)
    assert u.num_nodes == 140 and u.node_text_feat.size(0) == 50 + 11 + 20 and u.edge_text_feat.size(0) == 9
    # a member's rows still address its own texts
    for g, s in zip(gs, ptr[:-1].tolist()):
        assert torch.equal(u.node_text_feat[u.x[s:s + g.num_nodes]], g.node_text_feat[g.x])
    # no edge crosses a member boundary
    member = torch.bucketize(u.edge_index, ptr[1:], right=True)
    assert torch.equal(member[0], member[1])


def test_weighted_seed_list_counts():
    from stem_gnn_amd.data.multi import get_train_node_idx, mix_weights
    ptr = torch.tensor([0, 100, 350, 1000, 1030])
    w = [5, 10, 0.1, 1.5]
    got = get_train_node_idx(ptr, w, generator=torch.Generator().manual_seed(3))
    ref = O.get_train_node_idx(ptr, w, generator=torch.Generator().manual_seed(3))
    assert got.numel() == ref.numel() == 5 * 100 + 10 * 250 + 65 + 30 + 15
    cnt = torch.bincount(got, minlength=1030)
    assert bool((cnt[:100] == 5).all()) and bool((cnt[100:350] == 10).all())
    assert int(cnt[350:1000].sum()) == 65 and int(cnt[350:1000].max()) == 1           # a random 10 % once
    assert sorted(cnt[1000:].tolist()) == [1] * 15 + [2] * 15                        # all once, a random half twice
    assert torch.equal(got, ref)  # same generator, same draws: the restatement and the product agree entry by entry
    # integer weights need no randomness: members in order, each repeated whole
    assert torch.equal(get_train_node_idx(ptr[:3], [2, 1]), torch.cat([torch.arange(100).repeat(2), torch.arange(100, 350)]))
    assert list(mix_weights("all").values()) == [5, 5, 5, 5, 5, 10, 1, 0.1, 0.1]
    assert list(mix_weights("wo_arxiv")) == ["cora", "pubmed", "wikics", "WN18RR", "FB15K237", "chemhiv", "chemblpre", "chempcba"]
