"""gim_fine_fused (the whole fine level in one kernel, bf16 operand mode) against
  * the unfused bf16 launch sequence (same rounding points: agreement far below bf16 resolution of the features), and
  * the fp32 CPU oracle on the same bf16-valued fine maps (fine_preprocess.py:40-47, transformer.py:35-101,
    attentions.py:20-47, fine_matching.py:43-74) at bf16 tolerance,
on random matches that include image-border cells (zero-padded windows), a match count that is not a multiple of the
4 matches per workgroup, per-image scales, and M = 1."""
import pytest
import torch

import loftr_oracle as O
from tools import synth_loftr as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["bf16", "fp16"])   # the two 16-bit flavours of the kernel (csrc/gim_common.h)
def model_sd(request):
    model, sd = S.synthetic_model(request.param)
    return model.to("cuda:0"), sd


def _t(model):
    return torch.float16 if model.precision == "fp16" else torch.bfloat16


def _case(M, seed, hc=12, wc=16, bs=2, tdt=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    f0 = torch.randn(bs, 4 * hc, 4 * wc, 128, generator=g).to(tdt)
    f1 = torch.randn(bs, 4 * hc, 4 * wc, 128, generator=g).to(tdt)
    b = torch.randint(0, bs, (M,), generator=g)
    i = torch.randint(0, hc * wc, (M,), generator=g)
    j = torch.randint(0, hc * wc, (M,), generator=g)
    if M >= 4:  # corners / edges: windows hang over the map (F.unfold padding = zeros)
        i[0], j[0] = 0, hc * wc - 1
        i[1], j[1] = wc - 1, (hc - 1) * wc
        i[2], j[2] = 5, 5 * wc
    order = torch.argsort(b * hc * wc + i, stable=True)
    b, i, j = b[order], i[order], j[order]
    mk1c = torch.stack([(j % wc).float() * 8, (j // wc).float() * 8], 1)
    return f0, f1, b, i, j, mk1c


def _run(model, case, fused, scale1=None, debug=True):
    f0, f1, b, i, j, mk1c = [t.cuda() for t in case]
    model.debug = {} if debug else None
    try:
        out = model._fine_level(f0, f1, b, i, j, mk1c, scale1, scale1 is not None, (12, 16), (12, 16), (96, 128), fused)
        torch.cuda.synchronize()
    finally:
        model.debug = None
    return [None if t is None else t.float().cpu() for t in out]


def _oracle(sd, case, scale1=None):
    f0, f1, b, i, j, mk1c = case
    n0, n1 = f0.float().permute(0, 3, 1, 2).contiguous(), f1.float().permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        w0, w1 = O.fine_preprocess(n0, n1, b, i, j, (12, 16), (48, 64), 5)
        t0, t1 = O.local_feature_transformer(sd, "loftr_fine", w0, w1, 8, 1)
        fm = O.fine_matching(t0, t1, mk1c, mk1c, b, b.numel(), (96, 128), (48, 64), scale1, scale1 is not None)
    return t0, t1, fm


@pytest.mark.parametrize("M,seed", [(37, 1), (256, 2), (1, 3), (5, 4)])
def test_fused_matches_unfused_and_oracle(model_sd, M, seed):
    model, sd = model_sd
    case = _case(M, seed, tdt=_t(model))
    e_f, k_f, a0, a1 = _run(model, case, True)
    e_u, k_u, u0, u1 = _run(model, case, False)
    t0, t1, fm = _oracle(sd, case)
    scale = t0.abs().max().item()
    # fused vs unfused bf16: same operand roundings; only fp32 summation order, the bf16 KV operand of the attention and
    # isolated bf16 re-rounding flips differ
    for got, ref, nm in ((a0, u0, "fine0"), (a1, u1, "fine1")):
        err = (got - ref).abs()
        assert err.max().item() < 4e-2 * scale and err.mean().item() < 2e-3 * scale, (nm, err.max().item() / scale, err.mean().item() / scale)
    assert (e_f - e_u).abs().max() < 2e-2 and (k_f - k_u).abs().max() < 8e-2
    # fused vs fp32 oracle at bf16 tolerance -- and no worse than the unfused path is
    for got, unf, ref, nm in ((a0, u0, t0, "fine0"), (a1, u1, t1, "fine1")):
        ef, eu = (got - ref).abs().mean().item() / scale, (unf - ref).abs().mean().item() / scale
        assert ef < 6e-3 and ef < 1.5 * eu + 1e-4, (nm, ef, eu)
        assert (got - ref).abs().max().item() < 6e-2 * scale, nm
    assert (e_f - fm["expec_f"]).abs().max() < 4e-2
    assert (k_f - fm["mkpts1_f"]).abs().max() < 0.16   # 4 px per unit of expectation
    assert fm["expec_f"][:, :2].abs().max() > 0.02 or M == 1


def test_fused_with_scales_and_no_debug(model_sd):
    model, sd = model_sd
    case = _case(64, 7, tdt=_t(model))
    scale1 = torch.tensor([[1.5, 0.75], [2.0, 1.25]])
    e_f, k_f, d0, d1 = _run(model, case, True, scale1.cuda(), debug=False)
    assert d0 is None and d1 is None
    _, _, fm = _oracle(sd, case, scale1)
    assert (e_f - fm["expec_f"]).abs().max() < 4e-2
    assert (k_f - fm["mkpts1_f"]).abs().max() < 0.16 * 2.0


def test_end_to_end_bf16_uses_fused_and_agrees_with_unfused(model_sd):
    model, _ = model_sd
    c0, c1 = S.textured_pairs(2, 192, 256, seed=9)
    outs = {}
    for fused in (True, False):
        model.fine_fused = fused
        d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
        model(d)
        outs[fused] = {k: d[k].float().cpu() for k in ("b_ids", "i_ids", "j_ids", "expec_f", "mkpts1_f", "mkpts0_f")}
    model.fine_fused = True
    a, b = outs[True], outs[False]
    assert a["b_ids"].numel() > 300
    for k in ("b_ids", "i_ids", "j_ids", "mkpts0_f"):
        assert torch.equal(a[k], b[k]), k
    assert (a["expec_f"] - b["expec_f"]).abs().max() < 2e-2
    assert (a["mkpts1_f"] - b["mkpts1_f"]).abs().max() < 8e-2


def test_device_side_match_count_equals_host_side(model_sd):
    """gim_fine_fused_dev (launch over the capacity of the match lists, count read on the device) = the launch that knows the count:
    bit-identical outputs end to end, incl. the forward without matches (the count is zero, every workgroup leaves at once)"""
    model, _ = model_sd
    c0, c1 = S.textured_pairs(2, 192, 256, seed=11)
    outs = {}
    try:
        for dc in (True, False):
            model.fine_dev_count = dc
            d = {"image0": c0[:, :1].cuda(), "image1": c1[:, :1].cuda(), "color0": c0.cuda(), "color1": c1.cuda()}
            model(d)
            outs[dc] = {k: d[k].clone() for k in ("b_ids", "i_ids", "j_ids", "mconf", "expec_f", "mkpts0_f", "mkpts1_f")}
        assert outs[True]["b_ids"].numel() > 300
        for k, v in outs[True].items():
            assert v.shape == outs[False][k].shape and torch.equal(v, outs[False][k]), k
        model.fine_dev_count = True
        z = torch.zeros(1, 1, 192, 256).cuda()   # constant images: no coarse match clears the threshold
        d = {"image0": z, "image1": z, "color0": z.expand(-1, 3, -1, -1).contiguous(), "color1": z.expand(-1, 3, -1, -1).contiguous()}
        model(d)
        if d["b_ids"].numel() == 0:
            assert d["expec_f"].shape == (0, 3) and d["mkpts1_f"].shape == (0, 2)
    finally:
        model.fine_dev_count = True
