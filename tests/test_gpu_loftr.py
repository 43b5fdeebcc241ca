"""GPU end-to-end parity of gim_amd.loftr.LoFTR (HIP path through the C ABI) against the CPU oracle
(oracle/loftr_oracle.py, pinned to the reference) on the same seeded weights and inputs, plus
size-independent properties at the BASELINE size (640x480, batch 8)."""
import os

import numpy as np
import pytest
import torch

import loftr_oracle as O

pytestmark = pytest.mark.gpu


def _model(precision, sd):
    from gim_amd.loftr import LoFTR, get_cfg_defaults, lower_config
    cfg = lower_config(get_cfg_defaults())["loftr"]
    cfg["precision"] = precision
    m = LoFTR(cfg)
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    return m.eval().to("cuda:0")


def _data(c0, c1, dev=None, **extra):
    d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1, **extra}
    if dev is not None:
        d = {k: v.to(dev) for k, v in d.items()}
    return d


def _rel(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item(), \
        ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def _nchw(t, C):
    from gim_amd import ops
    return ops.nhwc_to_nchw(t.contiguous(), C).cpu()


@pytest.mark.parametrize("hw", [(64, 96), (96, 128)])
def test_fp32_end_to_end_matches_oracle(oracle_sd, hw):
    """fp32 mode: stage outputs within 1e-4, match indices exact, coordinates / confidences 1e-4."""
    dev = torch.device("cuda:0")
    m = _model("fp32", oracle_sd)
    m.debug = {}
    c0, c1 = O.seeded_images(2, *hw, seed=51)
    d = _data(c0, c1, dev)
    assert m(d) is None
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = O.loftr_forward(oracle_sd, _data(c0, c1))
        rc, rf = O.backbone(oracle_sd, torch.cat([c0, c1], 0))
    dbg = m.debug
    gc = torch.cat([_nchw(dbg["c0"], 256), _nchw(dbg["c1"], 256)], 0)
    gf = torch.cat([_nchw(dbg["f0"], 128), _nchw(dbg["f1"], 128)], 0)
    assert _rel(gc, rc)[1] < 1e-4, ("coarse map", _rel(gc, rc))
    assert _rel(gf, rf)[1] < 1e-4, ("fine map", _rel(gf, rf))
    conf = d["conf_matrix"].get().cpu()
    assert _rel(conf, ref["conf_matrix"])[1] < 1e-3, ("conf", _rel(conf, ref["conf_matrix"]))
    # reference key order / dtypes
    assert [k for k in d.keys()] == [k for k in ref.keys()]
    for k in ("b_ids", "i_ids", "j_ids", "m_bids"):
        assert d[k].dtype == torch.int64 and torch.equal(d[k].cpu(), ref[k]), k
    assert d["gt_mask"].dtype == torch.bool
    for k in ("mkpts0_c", "mkpts1_c", "mconf", "expec_f", "mkpts0_f", "mkpts1_f"):
        assert d[k].shape == ref[k].shape, k
        if ref[k].numel():
            assert (d[k].cpu() - ref[k]).abs().max() <= 1e-4 * max(1.0, ref[k].abs().max().item()), k
    assert d["hw0_c"] == ref["hw0_c"] and d["hw0_f"] == ref["hw0_f"] and d["W"] == 5 and d["bs"] == 2


def test_fp32_matches_golden_fixture(oracle_sd, golden_dir):
    """same, against the golden vectors recorded from the reference itself (tests/golden/e2e_64x96.npz)"""
    g = np.load(os.path.join(golden_dir, "e2e_64x96.npz"))
    m = _model("fp32", oracle_sd)
    c0, c1 = O.seeded_images(2, *g["hw"], seed=int(g["seed"]))
    d = _data(c0, c1, "cuda:0")
    m(d)
    for k in ("b_ids", "i_ids", "j_ids"):
        assert torch.equal(d[k].cpu(), torch.as_tensor(g[k])), k
    for k in ("mconf", "mkpts0_f", "mkpts1_f", "expec_f"):
        ref = torch.as_tensor(g[k])
        if ref.numel():
            assert (d[k].cpu() - ref).abs().max() <= 1e-4 * max(1.0, ref.abs().max().item()), k
    conf = d["conf_matrix"].get().cpu()
    assert (conf.max(dim=2)[0] - torch.as_tensor(g["conf_rowmax"])).abs().max() < 1e-5


def test_fp32_transformer_and_fine_stages_match_oracle(oracle_sd):
    """coarse transformer on random tokens (8 layers) and the fine path on planted matches, fp32, 1e-4"""
    from gim_amd import ops
    dev = torch.device("cuda:0")
    m = _model("fp32", oracle_sd)
    P = m._prepack(dev)
    g = torch.Generator().manual_seed(21)
    bs, L, C = 2, 48, 256
    f0, f1 = torch.randn(bs, L, C, generator=g), torch.randn(bs, L, C, generator=g)
    T = m._TfBuffers(2 * bs * L, C, torch.float32, dev)
    T.X32.copy_(torch.cat([f0, f1], 0).reshape(-1, C))
    T.CAT[:, :C].copy_(T.X32)
    m._transformer(P, "c", m.loftr_coarse, T, bs, L, bs, L)
    torch.cuda.synchronize()
    with torch.no_grad():
        r0, r1 = O.local_feature_transformer(oracle_sd, "loftr_coarse", f0, f1, 8, 4)
    got = T.X32.cpu().view(2 * bs, L, C)
    assert _rel(got[:bs], r0)[1] < 1e-4 and _rel(got[bs:], r1)[1] < 1e-4, (_rel(got[:bs], r0), _rel(got[bs:], r1))


@pytest.mark.parametrize("hw", [(96, 128)])
def test_bf16_end_to_end_close_to_oracle(oracle_sd, hw):
    """bf16 throughput mode: feature maps stay within bf16-level relative error of the fp32 oracle."""
    m = _model("bf16", oracle_sd)
    m.debug = {}
    c0, c1 = O.seeded_images(2, *hw, seed=51)
    d = _data(c0, c1, "cuda:0")
    m(d)
    torch.cuda.synchronize()
    with torch.no_grad():
        rc, rf = O.backbone(oracle_sd, torch.cat([c0, c1], 0))
    gc = torch.cat([_nchw(m.debug["c0"], 256), _nchw(m.debug["c1"], 256)], 0)
    gf = torch.cat([_nchw(m.debug["f0"], 128), _nchw(m.debug["f1"], 128)], 0)
    ec, ef = _rel(gc, rc), _rel(gf, rf)
    print("bf16 backbone rel err (fro, max): coarse", ec, "fine", ef)
    assert ec[0] < 3e-2 and ef[0] < 3e-2, (ec, ef)
    with torch.no_grad():
        ref = O.loftr_forward(oracle_sd, _data(c0, c1))
    conf = d["conf_matrix"].get().cpu()
    print("bf16 conf rel err", _rel(conf, ref["conf_matrix"]))
    assert torch.isfinite(conf).all()


def test_different_image_shapes(oracle_sd):
    """hw0 != hw1 takes the two-backbone-call branch (loftr.py:62-63)"""
    m = _model("fp32", oracle_sd)
    g = torch.Generator().manual_seed(5)
    c0, c1 = torch.rand(1, 3, 64, 96, generator=g), torch.rand(1, 3, 96, 64, generator=g)
    d = _data(c0, c1, "cuda:0")
    m(d)
    with torch.no_grad():
        ref = O.loftr_forward(oracle_sd, _data(c0, c1))
    for k in ("b_ids", "i_ids", "j_ids"):
        assert torch.equal(d[k].cpu(), ref[k]), k
    conf = d["conf_matrix"].get().cpu()
    assert _rel(conf, ref["conf_matrix"])[1] < 1e-3


def test_full_size_properties_bf16(oracle_sd):
    """BASELINE config 2 shape (640x480, batch 8 pairs): no oracle at this size (tens of seconds per pair
    on CPU); check size-independent properties instead: determinism, ordering, mutual uniqueness,
    border rule, thresholds, index ranges."""
    m = _model("bf16", oracle_sd)
    c0, c1 = O.seeded_images(8, 480, 640, seed=1234)
    d = _data(c0, c1, "cuda:0")
    m(d)
    M = d["b_ids"].numel()
    b, i, j, conf = d["b_ids"].cpu(), d["i_ids"].cpu(), d["j_ids"].cpu(), d["mconf"].cpu()
    assert d["hw0_c"] == torch.Size([60, 80]) and d["hw0_f"] == torch.Size([240, 320])
    key = b * 4800 + i
    assert (key[1:] > key[:-1]).all() if M > 1 else True          # torch.where order, one match per (b,i)
    assert torch.unique(b * 4800 + j).numel() == M                  # mutual: one match per (b,j)
    assert (conf > 0.2).all() and (conf <= 1.0).all()
    for idx in (i, j):                                              # border_rm = 2 on all four axes
        y, x = idx // 80, idx % 80
        assert ((y >= 2) & (y < 58) & (x >= 2) & (x < 78)).all()
    assert torch.equal(d["mkpts0_f"].cpu(), torch.stack([i % 80, i // 80], 1).float() * 8)
    assert ((d["mkpts1_f"].cpu() - d["mkpts1_c"].cpu()).abs() <= 4.0 + 1e-4).all()  # |coords| <= 1, * 2 * 2
    d2 = _data(c0, c1, "cuda:0")
    m(d2)
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts1_f"):
        assert torch.equal(d[k], d2[k]), f"non-deterministic {k}"
    # (batch composition -- pair b alone == pair b inside the batch -- is asserted bit for bit in the fp32 mode below.  In the 16-bit modes
    # the launch shapes differ with the batch: layer 3's 9 600 pixel rows of a single pair are not a multiple of the fused tail's 256-row
    # tile, so a single pair runs the unfused convolutions, and tile-count thresholds pick other kernels -- other fp32 summation orders
    # in front of a 16-bit rounding.  With THIS model's uncalibrated random weights the similarities saturate (confidences of 1.0) and
    # a difference of one 16-bit ulp in a feature can move the arg-max between two rows: the 1-2 matches per pair of this workload are
    # no basis for an assertion; the calibrated workload's flip rates are bounded in test_gpu_loftr_fullsize.py.)


def test_batch_composition_invariance_fp32(oracle_sd):
    """fp32 mode, 640x480: pair b alone == pair b inside a batch of 4, bit for bit (every GEMM runs the same K order whatever the tile
    shape the launch picks)"""
    m = _model("fp32", oracle_sd)
    c0, c1 = O.seeded_images(4, 480, 640, seed=1234)
    d = _data(c0, c1, "cuda:0")
    m(d)
    b = d["b_ids"].cpu()
    for pb in (0, 3):
        d1 = _data(c0[pb:pb + 1], c1[pb:pb + 1], "cuda:0")
        m(d1)
        sel = b == pb
        for k in ("i_ids", "j_ids", "mconf", "mkpts1_f"):
            assert torch.equal(d1[k].cpu(), d[k].cpu()[sel]), (pb, k)


def _pad_mask(n, h, w, valid):
    m = torch.zeros(n, h, w, dtype=torch.bool)
    for b, (vh, vw) in enumerate(valid):
        m[b, :vh, :vw] = True
    return m


def test_fp32_masked_stages_match_oracle(oracle_sd):
    """padding masks through the HIP path: masked coarse transformer and masked coarse matching"""
    from gim_amd import ops
    dev = torch.device("cuda:0")
    m = _model("fp32", oracle_sd)
    P = m._prepack(dev)
    hw_c = (12, 16)
    m0, m1 = _pad_mask(2, *hw_c, [(12, 13), (9, 16)]), _pad_mask(2, *hw_c, [(10, 16), (12, 11)])
    g = torch.Generator().manual_seed(61)
    bs, L, C = 2, 192, 256
    f0, f1 = torch.randn(bs, L, C, generator=g), torch.randn(bs, L, C, generator=g)
    T = m._TfBuffers(2 * bs * L, C, torch.float32, dev)
    T.X32.copy_(torch.cat([f0, f1], 0).reshape(-1, C))
    T.CAT[:, :C].copy_(T.X32)
    T.MASK = torch.cat([m0.reshape(-1), m1.reshape(-1)]).to(torch.uint8).to(dev)
    m._transformer(P, "c", m.loftr_coarse, T, bs, L, bs, L)
    torch.cuda.synchronize()
    with torch.no_grad():
        r0, r1 = O.local_feature_transformer(oracle_sd, "loftr_coarse", f0, f1, 8, 4, m0.flatten(-2), m1.flatten(-2))
    got = T.X32.cpu().view(2 * bs, L, C)
    assert _rel(got[:bs], r0)[1] < 1e-4 and _rel(got[bs:], r1)[1] < 1e-4, (_rel(got[:bs], r0), _rel(got[bs:], r1))
    # masked coarse matching on planted features
    pf0, pf1, _ = O.planted_coarse_features(2, hw_c, sigma=1.0, eps=0.5, seed=31)
    conf = O.conf_matrix_dual_softmax(pf0, pf1, 0.1, m0.flatten(-2), m1.flatten(-2))
    ref = O.get_coarse_match(conf, (96, 128), (96, 128), hw_c, hw_c, 0.2, 2, None, None, m0, m1)
    r = ops.coarse_match(pf0.to(dev), pf1.to(dev), hw_c, hw_c, 8.0, 0.1, 0.2, 2, None, None,
                         m0.reshape(2, -1).to(torch.uint8).to(dev).contiguous(),
                         m1.reshape(2, -1).to(torch.uint8).to(dev).contiguous())
    M = int(r.count[0])
    assert M == ref["b_ids"].numel() == 45
    for k in ("b_ids", "i_ids", "j_ids"):
        assert torch.equal(getattr(r, k)[:M].cpu(), ref[k]), k
    assert (r.mconf[:M].cpu() - ref["mconf"]).abs().max() < 1e-5
    cm = ops.coarse_conf_matrix(r).cpu()
    assert (cm - conf).abs().max() < 1e-5


@pytest.mark.parametrize("graph", [False, True])
def test_fp32_end_to_end_masked(oracle_sd, graph):
    m = _model("fp32", oracle_sd)
    m.use_graph = graph
    c0, c1 = O.seeded_images(2, 64, 96, seed=71)
    m0, m1 = _pad_mask(2, 8, 12, [(8, 9), (6, 12)]), _pad_mask(2, 8, 12, [(7, 12), (8, 10)])
    c0 = c0 * torch.nn.functional.interpolate(m0[:, None].float(), scale_factor=8)
    c1 = c1 * torch.nn.functional.interpolate(m1[:, None].float(), scale_factor=8)
    d = _data(c0, c1, "cuda:0", mask0=m0, mask1=m1)
    m(d)
    if graph:  # a shape is captured the second time it is seen; the third call replays
        for _ in range(2):
            d = _data(c0, c1, "cuda:0", mask0=m0, mask1=m1)
            m(d)
        assert len(m._graphs) == 1
    with torch.no_grad():
        ref = O.loftr_forward(oracle_sd, _data(c0, c1, mask0=m0, mask1=m1))
    for k in ("b_ids", "i_ids", "j_ids"):
        assert torch.equal(d[k].cpu(), ref[k]), k
    conf = d["conf_matrix"].get().cpu()
    assert _rel(conf, ref["conf_matrix"])[1] < 1e-3
    for k in ("mconf", "mkpts0_f", "mkpts1_f"):
        if ref[k].numel():
            assert (d[k].cpu() - ref[k]).abs().max() <= 1e-4 * max(1.0, ref[k].abs().max().item()), k
