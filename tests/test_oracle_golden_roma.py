"""CPU: oracle/roma_oracle.py replayed against the vectors oracle/make_golden_roma.py recorded from the reference's own
RoMa (networks/roma/roma.py, networks/roma/dino.py) -- SURVEY 8a row a14."""
import os

import numpy as np
import pytest
import torch

import dkm_oracle as DO
import roma_oracle as O


def _close(a, b, tol=1e-4):
    a, b = torch.as_tensor(np.asarray(a)), torch.as_tensor(np.asarray(b))
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, b.abs().max().item())
    assert (a - b).abs().max().item() <= tol * scale


def test_param_specs():
    spec, dspec = O.roma_param_spec(), O.dino_param_spec()
    assert len(spec) == 603
    assert sum(int(np.prod(v)) for v in dspec.values()) == 304368640               # DINOv2 ViT-L/14 (with mask token)
    assert spec["decoder.embedding_decoder.to_out.weight"] == (4097, 1024)
    assert spec["decoder.conv_refiner.1.block1.0.weight"] == (24, 1, 5, 5)          # 2*9 + 6 = 24 input channels, depthwise


def _inputs(golden_dir):
    g = np.load(os.path.join(golden_dir, "roma_stages.npz"))
    H, W = (int(v) for v in g["hw"])
    im0, im1 = DO.seeded_pair(*(int(v) for v in g["image_hw"]), int(g["seed"]))
    up = lambda t, s: torch.nn.functional.interpolate(t, size=s, mode="bilinear", align_corners=False)  # noqa: E731
    return g, H, W, im0, im1, up


def _decoder_stages(sd, g, dino16):
    a = O._proj(sd, "16", dino16)
    c = torch.cat((a.chunk(2)[1], a.chunk(2)[0]))
    _close(O.gp_forward(sd, a, c), g["gp"], 1e-3)
    cls, cert = O.transformer_decoder(sd, torch.as_tensor(g["gp"]), a)
    _close(cls[:, ::64], g["cls_sub"], 1e-3)
    _close(cert, g["gm_certainty"], 1e-3)
    _close(O.cls_to_flow_refine(cls), g["gm_flow"], 1e-3)


def test_stage_goldens(golden_dir):
    """VGG pyramid, proj, GP, transformer match decoder and anchor regression against the reference's recorded tensors; the
    scale-16 input is the reference's own recorded DINOv2 output (the 24-block ViT-L replay is the slow test below)"""
    g, H, W, im0, im1, up = _inputs(golden_dir)
    sd = O.make_roma_state_dict(0)
    with torch.no_grad():
        pyr = O.vgg_pyramid(sd, torch.cat((up(im0, (H, W)), up(im1, (H, W)))))
        _close(pyr[8][:, ::8], g["vgg8_sub"])
        _close(pyr[1][:, ::16, ::4, ::4], g["vgg1_sub"])
        _decoder_stages(sd, g, torch.as_tensor(g["dino16"]))


@pytest.mark.skipif(not os.environ.get("GIM_SLOW_TESTS"), reason="DINOv2 ViT-L replay + whole pipeline (3 ViT-L passes on the CPU): "
                    "GIM_SLOW_TESTS=1; oracle/make_golden_roma.py asserts the same comparisons when it records the vectors")
def test_dino_and_match_golden(golden_dir):
    g, H, W, im0, im1, up = _inputs(golden_dir)
    gm = np.load(os.path.join(golden_dir, "roma_match.npz"))
    sd, dsd = O.make_state_dicts(0)
    with torch.no_grad():
        pyr = O.encoder(sd, dsd, torch.cat((up(im0, (H, W)), up(im1, (H, W)))))
        _close(pyr[16], g["dino16"], 2e-4)
        _decoder_stages(sd, g, pyr[16])
        up_res = tuple(int(v) for v in gm["up"])
        warp, cc = O.match(sd, dsd, im0, im1, H, W, up_res)
    _close(warp[::2, ::2], gm["warp"], 2e-3)
    _close(cc[::2, ::2], gm["certainty"], 5e-3)
    assert warp.shape == (up_res[0], 2 * up_res[1], 4) and warp.abs().max() <= 1
