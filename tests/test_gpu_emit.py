"""Output hand-out kernels (emit.hip) and the HIP-graph bookkeeping of LoFTR.forward around them."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_copy_segments_and_fill():
    from gim_amd import ops
    g = torch.Generator().manual_seed(0)
    srcs = [torch.randint(-2 ** 40, 2 ** 40, (n,), generator=g).cuda() for n in (1, 7, 1000, 12345)]
    srcs += [torch.randn(n, 2, generator=g).cuda() for n in (3, 4097)]
    srcs += [torch.randint(0, 255, (n,), generator=g, dtype=torch.uint8).cuda() for n in (5, 33, 100001)]   # odd byte counts
    big = torch.randn(1 << 16, generator=g).cuda()
    srcs += [big[1:1 + 4099]]                                                                          # 4-byte aligned only
    dsts = [torch.full_like(s, 77).contiguous() for s in srcs]
    fills = [torch.ones(n, dtype=torch.bool).cuda() for n in (1, 13, 5000)]
    pairs = list(zip([s.contiguous() if s.is_contiguous() else s for s in srcs], dsts)) + [(None, f) for f in fills] + \
        [(torch.empty(0).cuda(), torch.empty(0).cuda())]
    ops.copy_segments(pairs)     # 14 segments: two launches
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)
    for f in fills:
        assert not f.any()


def test_pack_matches_kernel_vs_torch():
    from gim_amd.runner import pack_matches
    g = torch.Generator().manual_seed(1)
    for M in (0, 1, 257, 11000):
        d = {"mkpts0_f": torch.rand(M, 2, generator=g) * 640, "mkpts1_f": torch.rand(M, 2, generator=g) * 480,
             "mconf": torch.rand(M, generator=g), "m_bids": torch.randint(0, 8, (M,), generator=g)}
        dc = {k: v.cuda() for k, v in d.items()}
        for ids in ([40 + b for b in range(8)], [3, 9, 1, 0, 200, 7, 7, 12], 40):
            want = pack_matches(d, ids)
            got = pack_matches(dc, ids)
            assert got.shape == (M, 6) and torch.equal(got.cpu(), want)


def test_graph_survives_precision_round_trip_and_reload():
    """bf16 forward, set_precision('fp32'), back to 'bf16', forward of the SAME shape -- and a forward after
    load_state_dict on a seen shape -- must capture cleanly (weights are packed before capture starts, the seen-shape
    counters are dropped with the graphs) and reproduce the eager result."""
    import warnings
    from tools import synth_loftr as S
    model, sd = S.synthetic_model("bf16")
    model = model.cuda()
    c0, c1 = S.textured_pairs(1, 96, 128, seed=5, frac=1.0)

    def fwd():
        d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
        model(d)
        return d

    with warnings.catch_warnings():
        warnings.simplefilter("error")          # a failed capture warns and falls back to eager launches
        ref = fwd()                              # eager (first sighting)
        a = fwd()                                # captured
        assert len(model._graphs) == 1
        model.set_precision("fp32")
        assert not model._graphs and not model._seen and model._packed is None
        model.set_precision("bf16")
        b = fwd()                                # eager again: nothing is packed, the counters are gone
        assert not model._graphs
        c = fwd()                                # capture with a cold pack cache inside _coarse_stage_graphed
        assert len(model._graphs) == 1 and model.use_graph
        model.load_state_dict({k: v.clone() for k, v in sd.items()})
        model._seen[model._graph_key(c0.cuda(), c1.cuda(), None, None)] = 1   # a stale counter, as before the fix: straight into capture
        e = fwd()
        assert model.use_graph
    for d in (a, b, c, e):
        for k in ("b_ids", "i_ids", "j_ids", "m_bids"):
            assert torch.equal(d[k], ref[k]), k
        for k in ("mconf", "mkpts0_f", "mkpts1_f", "mkpts0_c", "mkpts1_c"):
            assert torch.equal(d[k], ref[k]), k
        assert d["gt_mask"].dtype == torch.bool and d["gt_mask"].numel() == d["b_ids"].numel() and not d["gt_mask"].any()
    assert ref["b_ids"].numel() > 20
    # graph replays hand out PRIVATE copies: the previous call's lists survive the next replay
    keep = {k: a[k].clone() for k in ("b_ids", "mconf", "mkpts1_c")}
    fwd()
    for k, v in keep.items():
        assert torch.equal(a[k], v)
