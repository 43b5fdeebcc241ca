"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, the LoFTR shell
keeps the reference's checkpoint surface, and the weight packing / K-group table are consistent with
the kernel's gather (checked by a pure-torch emulation of the implicit GEMM addressing).
No compute call into the HIP library happens here (no GPU)."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

import loftr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gim_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gim_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gim_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    for name in declared:
        assert hasattr(_lib.lib, name), name
    assert _lib.lib.gim_version() >= 100
    assert _lib.lib.gim_ktile_bytes() == 128


def test_no_cpu_fallback():
    from gim_amd import _lib, ops
    with pytest.raises(_lib.GimHipError):
        ops.upsample2x_add(torch.zeros(1, 2, 2, 4), torch.zeros(1, 4, 4, 4))
    from gim_amd.loftr import LoFTR, get_cfg_defaults, lower_config
    m = LoFTR(lower_config(get_cfg_defaults())["loftr"])
    x = torch.zeros(1, 3, 32, 32)
    with pytest.raises(_lib.GimHipError):
        m({"image0": x[:, :1], "image1": x[:, :1], "color0": x, "color1": x})


def test_state_dict_surface_matches_reference(oracle_sd):
    from gim_amd.loftr import LoFTR, get_cfg_defaults, lower_config
    cfg = lower_config(get_cfg_defaults())["loftr"]
    assert cfg == {k: v for k, v in cfg.items()}  # plain dict
    for k, v in O.DEFAULT_CONFIG.items():  # same effective config as the reference's defaults
        assert cfg[k] == v, k
    m = LoFTR(cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(oracle_sd.keys())
    for k in sd:
        assert sd[k].shape == oracle_sd[k].shape and sd[k].dtype == oracle_sd[k].dtype, k
    # prefix stripping of loftr.py:93-99
    pref = {("model." + k if i % 2 else "matcher." + k): v.clone() for i, (k, v) in enumerate(oracle_sd.items())}
    assert not m.load_state_dict(pref, strict=True).missing_keys
    assert torch.equal(m.state_dict()["loftr_fine.layers.1.mlp.2.weight"], oracle_sd["loftr_fine.layers.1.mlp.2.weight"])


def test_posenc_table_matches_oracle():
    from gim_amd.loftr import LoFTR, get_cfg_defaults, lower_config
    m = LoFTR(lower_config(get_cfg_defaults())["loftr"])
    pe = m._pos_encoding(256, 60, 80, torch.device("cpu"))
    ref = O.position_encoding(256, 60, 80)[0].permute(1, 2, 0).reshape(4800, 256)
    assert torch.equal(pe, ref)


def _emulate_igemm(x_nhwc, pk, stride, pad):
    """What the kernel computes, restated with the packed operands: for every 16-byte K group g the table
    gives (dy, dx, c); A[m, g*GE:(g+1)*GE] = x[b, ho*s-p+dy, wo*s-p+dx, c:c+GE] or 0 outside the image."""
    from gim_amd.packing import group_elems
    B, H, W, cs = x_nhwc.shape
    ge = group_elems(pk.dtype)
    Ho, Wo = (H + 2 * pad - pk.kh) // stride + 1, (W + 2 * pad - pk.kw) // stride + 1
    ngrp = pk.kpad // ge
    A = torch.zeros(B, Ho, Wo, pk.kpad)
    tab = pk.ktab.cpu().to(torch.int64) & 0xFFFFFFFF
    assert tab.numel() == (pk.kpad // (128 // (16 // ge)) + 2) * 8
    assert (tab[ngrp:] >> 24 == 255).all(), "look-ahead slabs must be marked invalid"
    for g in range(ngrp):
        e = int(tab[g])
        c, dx, dy = e & 0xFFFF, (e >> 16) & 0xFF, (e >> 24) & 0xFF
        if dy == 255:
            continue
        for ho in range(Ho):
            iy = ho * stride - pad + dy
            if not 0 <= iy < H:
                continue
            for wo in range(Wo):
                ix = wo * stride - pad + dx
                if 0 <= ix < W:
                    A[:, ho, wo, g * ge:(g + 1) * ge] = x_nhwc[:, iy, ix, c:c + ge]
    y = A.reshape(-1, pk.kpad) @ pk.w.float().cpu().t()
    if pk.bias is not None:
        y = y + pk.bias.cpu()
    return y.reshape(B, Ho, Wo, pk.npad)[..., :pk.n_store]


@pytest.mark.parametrize("case", [(3, 16, 7, 2), (12, 20, 3, 1), (12, 20, 3, 2), (20, 12, 1, 1), (8, 8, 1, 2)],
                         ids=str)
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_packing_and_ktab_against_conv2d(case, dt):
    from gim_amd import _lib
    from gim_amd.packing import cstore, pack_conv
    cin, cout, k, stride = case
    gd = _lib.GIM_BF16 if dt == "bf16" else _lib.GIM_F32
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(2, cin, 9, 11, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g)
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g), torch.randn(cout, generator=g),
          torch.rand(cout, generator=g) + 0.5, 1e-5)
    pk = pack_conv(w, bn, gd, "cpu", stride=stride, pad=k // 2, cin_pad=cstore(cin, gd))
    cs = pk.cin_pad
    xn = torch.zeros(2, 9, 11, cs)
    xn[..., :cin] = x.permute(0, 2, 3, 1)
    got = _emulate_igemm(xn, pk, stride, k // 2)
    ref = F.batch_norm(F.conv2d(x, w, stride=stride, padding=k // 2), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    ref = ref.permute(0, 2, 3, 1)
    tol = 3e-2 if dt == "bf16" else 1e-4  # bf16: packed weights are rounded
    assert (got[..., :cout] - ref).abs().max() <= tol * ref.abs().max()
    assert (got[..., cout:] == 0).all()
    assert pk.npad % 64 == 0 and pk.n_store % 4 == 0 and pk.n_store >= cout


# ---- gim_lightglue host logic ---------------------------------------------------------------------------------
def test_lightglue_state_dict_surface_and_qkv_permutation():
    """reference parameter names/shapes (24 + 251 tensors) and the Wqkv row permutation that turns the
    reference's interleaved (head, dim, {q,k,v}) output features (lightglue.py:147-148) into [q | k | v]"""
    import lightglue_oracle as LO
    from gim_amd.lightglue import LightGlue, SuperPoint
    sp_sd, lg_sd = LO.make_state_dicts(0)
    det = SuperPoint({"max_num_keypoints": 2048, "force_num_keypoints": True, "detection_threshold": 0.0, "nms_radius": 3,
                      "trainable": False})
    lg = LightGlue({"filter_threshold": 0.1, "flash": False, "checkpointed": True})
    assert {k: tuple(v.shape) for k, v in det.state_dict().items()} == {k: tuple(v.shape) for k, v in sp_sd.items()}
    assert {k: tuple(v.shape) for k, v in lg.state_dict().items()} == {k: tuple(v.shape) for k, v in lg_sd.items()}
    lg.load_state_dict(lg_sd)
    H, dh, d = 4, 64, 256
    perm = torch.empty(3 * d, dtype=torch.long)
    for s in range(3):
        for h in range(H):
            for j in range(dh):
                perm[s * d + h * dh + j] = h * 3 * dh + j * 3 + s
    x = torch.randn(5, d)
    w, b = lg_sd["transformers.0.self_attn.Wqkv.weight"], lg_sd["transformers.0.self_attn.Wqkv.bias"]
    ref = torch.nn.functional.linear(x, w, b).unflatten(-1, (H, -1, 3))            # [5, H, dh, 3]
    got = torch.nn.functional.linear(x, w[perm], b[perm])
    for s in range(3):
        assert torch.equal(got[:, s * d:(s + 1) * d].reshape(5, H, dh), ref[..., s])
    with pytest.raises(NotImplementedError):
        LightGlue({"depth_confidence": 0.95})


def test_precision_resolver(monkeypatch):
    """GIM_PRECISION=fp16 (gim_loftr's default spelling) must not silently put the secondary engines on the fp32 path"""
    from gim_amd.precision import resolve
    monkeypatch.delenv("GIM_PRECISION", raising=False)
    assert resolve(None, "loftr", default="fp16") == "fp16" and resolve(None, "gim_dkm") == "bf16"
    monkeypatch.setenv("GIM_PRECISION", "fp16")
    assert resolve(None, "loftr", default="fp16") == "fp16" and resolve(None, "gim_roma") == "bf16" and resolve(None, "SuperPoint") == "bf16"
    assert resolve("fp32", "LightGlue") == "fp32"
    assert resolve("fp16", "gim_dkm") == "fp16"     # round 5: an EXPLICIT fp16 selects the engines' IEEE-fp16 flavour
    with pytest.raises(ValueError):
        resolve("fp8", "gim_dkm")           # explicit request for a mode the engine does not have
    monkeypatch.setenv("GIM_PRECISION", "fp64")
    for eng in ("loftr", "gim_dkm"):
        with pytest.raises(ValueError):
            resolve(None, eng)
