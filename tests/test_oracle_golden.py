"""CPU: the oracle (oracle/loftr_oracle.py) replayed against the golden vectors that
oracle/make_golden.py recorded from the *reference's own modules* (see that script for file:line)."""
import os

import numpy as np
import torch

import loftr_oracle as O


def _close(a, b, tol=1e-5):
    a = torch.as_tensor(np.asarray(a))
    b = torch.as_tensor(np.asarray(b))
    assert a.shape == b.shape
    if a.dtype in (torch.int64, torch.bool):
        assert torch.equal(a, b)
    elif a.numel():
        scale = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() <= tol * scale


def test_state_dict_spec_matches_reference_count(oracle_sd):
    # 375 tensors / 16.18 M parameters (SURVEY section 5 "Checkpoint / resume")
    assert len(oracle_sd) == 375
    n = sum(v.numel() for k, v in oracle_sd.items() if not k.endswith("num_batches_tracked"))
    assert n == 16211853 - sum(1 for k in oracle_sd if k.endswith("num_batches_tracked"))


def test_backbone_golden(oracle_sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "backbone.npz"))
    c0, c1 = O.seeded_images(1, *g["hw"], seed=int(g["seed"]))
    with torch.no_grad():
        oc, of = O.backbone(oracle_sd, torch.cat([c0, c1], 0))
    _close(oc, g["coarse"])
    _close(of[:, :, ::4, ::4], g["fine_sub"])
    assert abs(of.double().sum().item() - float(g["fine_sum"])) <= 1e-5 * float(g["fine_abs"])


def test_posenc_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "posenc.npz"))
    pe = O.position_encoding(256, 60, 80)
    _close(pe[0, :, :3, :5], g["pe_corner"], 0.0)
    _close(pe[0, :, 59, 79], g["pe_last"], 0.0)
    # the buggy divisor: div_term = exp(-2k)  
    assert abs(pe[0, 4, 0, 0].item() - np.sin(np.float32(np.exp(np.float32(-2.0))))) < 1e-7


def test_coarse_transformer_golden(oracle_sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "coarse_transformer.npz"))
    gen = torch.Generator().manual_seed(int(g["seed"]))
    f0 = torch.randn(*g["shape"], generator=gen)
    f1 = torch.randn(*g["shape"], generator=gen)
    with torch.no_grad():
        o0, o1 = O.local_feature_transformer(oracle_sd, "loftr_coarse", f0, f1, 8, 4)
    _close(o0, g["out0"])
    _close(o1, g["out1"])


def _coarse_case(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"coarse_match_{tag}.npz"))
    hw_c, hw_i = tuple(g["hw_c"]), tuple(g["hw_i"])
    f0, f1, _ = O.planted_coarse_features(2, hw_c, sigma=float(g["sigma"]), eps=float(g["eps"]),
                                          seed=int(g["seed"]))
    s0 = torch.as_tensor(g["scale0"]) if "scale0" in g else None
    s1 = torch.as_tensor(g["scale1"]) if "scale1" in g else None
    conf = O.conf_matrix_dual_softmax(f0, f1, 0.1)
    out = O.get_coarse_match(conf, hw_i, hw_i, hw_c, hw_c, 0.2, 2, s0, s1)
    return g, out


def test_coarse_match_golden(golden_dir):
    for tag in ("plain", "scaled"):
        g, out = _coarse_case(golden_dir, tag)
        assert out["b_ids"].numel() >= 50
        for k in ("b_ids", "i_ids", "j_ids", "m_bids"):
            assert out[k].dtype == torch.int64
            _close(out[k], g[k])
        for k in ("mkpts0_c", "mkpts1_c", "mconf"):
            _close(out[k], g[k], 1e-6)


def test_fine_golden(oracle_sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "fine.npz"))
    gc, cm = _coarse_case(golden_dir, "scaled")
    hw_c, hw_i, hw_f = tuple(gc["hw_c"]), tuple(gc["hw_i"]), tuple(g["hw_f"])
    gen = torch.Generator().manual_seed(int(g["seed"]))
    ff0 = torch.randn(2, 128, *hw_f, generator=gen)
    ff1 = torch.randn(2, 128, *hw_f, generator=gen)
    b_ids, i_ids, j_ids = (torch.as_tensor(g[k]) for k in ("b_ids", "i_ids", "j_ids"))
    u0, u1 = O.fine_preprocess(ff0, ff1, b_ids, i_ids, j_ids, hw_c, hw_f, 5)
    _close(u0[:4], g["unfold0_first"], 0.0)
    with torch.no_grad():
        t0, t1 = O.local_feature_transformer(oracle_sd, "loftr_fine", u0, u1, 8, 1)
    out = O.fine_matching(t0, t1, cm["mkpts0_c"], cm["mkpts1_c"], b_ids, len(cm["mconf"]), hw_i, hw_f,
                          torch.as_tensor(gc["scale1"]), True)
    for k in ("expec_f", "mkpts0_f", "mkpts1_f"):
        _close(out[k], g[k], 1e-5)


def test_end_to_end_golden(oracle_sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e_64x96.npz"))
    c0, c1 = O.seeded_images(2, *g["hw"], seed=int(g["seed"]))
    with torch.no_grad():
        d = O.loftr_forward(oracle_sd, {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1})
    for k in ("b_ids", "i_ids", "j_ids", "m_bids", "mkpts0_c", "mkpts1_c", "mconf", "expec_f",
              "mkpts0_f", "mkpts1_f"):
        _close(d[k], g[k], 1e-5)
    _close(d["conf_matrix"].max(dim=2)[0], g["conf_rowmax"], 1e-5)
    _close(d["conf_matrix"].max(dim=1)[0], g["conf_colmax"], 1e-5)
    # key insertion order of the reference's data dict
    ref_keys = [str(k) for k in g["key_order"]]
    assert [k for k in d.keys()] == ref_keys


def test_empty_match_case(oracle_sd):
    # zero matches must not raise and must give zero-length tensors (fine_preprocess.py:34-37,
    # fine_matching.py:33-41)
    conf = torch.zeros(1, 48, 48)
    out = O.get_coarse_match(conf, (48, 64), (48, 64), (6, 8), (6, 8))
    assert out["b_ids"].numel() == 0 and out["mkpts0_c"].shape == (0, 2)
    u0, u1 = O.fine_preprocess(torch.zeros(1, 128, 24, 32), torch.zeros(1, 128, 24, 32),
                               out["b_ids"], out["i_ids"], out["j_ids"], (6, 8), (24, 32))
    fm = O.fine_matching(u0, u1, out["mkpts0_c"], out["mkpts1_c"], out["b_ids"], 0, (48, 64), (24, 32))
    assert fm["expec_f"].shape == (0, 3) and fm["mkpts1_f"].shape == (0, 2)


def _pad_mask(n, h, w, valid):
    m = torch.zeros(n, h, w, dtype=torch.bool)
    for b, (vh, vw) in enumerate(valid):
        m[b, :int(vh), :int(vw)] = True
    return m


def test_masked_golden(oracle_sd, golden_dir):
    """padding masks: LinearAttention q/kv masks (attentions.py:35-39), masked_fill(-INF)
    (coarse_matching.py:116-117) and mask_border_with_padding (:29-44)"""
    g = np.load(os.path.join(golden_dir, "masked.npz"))
    hw_c, hw_i = tuple(g["hw_c"]), tuple(g["hw_i"])
    m0, m1 = _pad_mask(2, *hw_c, g["valid0"]), _pad_mask(2, *hw_c, g["valid1"])
    gen = torch.Generator().manual_seed(int(g["seed_tf"]))
    tf0, tf1 = torch.randn(2, 192, 256, generator=gen), torch.randn(2, 192, 256, generator=gen)
    with torch.no_grad():
        o0, o1 = O.local_feature_transformer(oracle_sd, "loftr_coarse", tf0, tf1, 8, 4, m0.flatten(-2), m1.flatten(-2))
    _close(o0[:, ::4], g["tf_out0_sub"])
    _close(o1[:, ::4], g["tf_out1_sub"])
    f0, f1, _ = O.planted_coarse_features(2, hw_c, sigma=1.0, eps=0.5, seed=int(g["seed_cm"]))
    conf = O.conf_matrix_dual_softmax(f0, f1, 0.1, m0.flatten(-2), m1.flatten(-2))
    out = O.get_coarse_match(conf, hw_i, hw_i, hw_c, hw_c, 0.2, 2, None, None, m0, m1)
    assert out["b_ids"].numel() == 45
    for k in ("b_ids", "i_ids", "j_ids"):
        _close(out[k], g[k])
    for k in ("mkpts0_c", "mkpts1_c", "mconf"):
        _close(out[k], g[k], 1e-6)


def test_end_to_end_masked_golden(oracle_sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "e2e_masked.npz"))
    c0, c1 = O.seeded_images(2, *g["hw"], seed=int(g["seed"]))
    m0, m1 = _pad_mask(2, 8, 12, g["valid0"]), _pad_mask(2, 8, 12, g["valid1"])
    c0 = c0 * torch.nn.functional.interpolate(m0[:, None].float(), scale_factor=8)
    c1 = c1 * torch.nn.functional.interpolate(m1[:, None].float(), scale_factor=8)
    with torch.no_grad():
        d = O.loftr_forward(oracle_sd, {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1,
                                        "mask0": m0, "mask1": m1})
    for k in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_f", "mkpts1_f"):
        _close(d[k], g[k], 1e-5)
    _close(d["conf_matrix"].max(dim=2)[0], g["conf_rowmax"], 1e-5)
