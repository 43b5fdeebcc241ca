"""CPU: oracle/dkm_oracle.py replayed against the vectors oracle/make_golden_dkm.py recorded from the reference's own
DKMv3 (networks/dkm/models/dkm.py, model_zoo/DKMv3.py) -- SURVEY 8a row a13.  The HIP path for this row is round 2;
the pinned oracle is its acceptance test."""
import os

import numpy as np
import torch
import torch.nn.functional as F

import dkm_oracle as O


def _close(a, b, tol=1e-4):
    a, b = torch.as_tensor(np.asarray(a)), torch.as_tensor(np.asarray(b))
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, b.abs().max().item())
    assert (a - b).abs().max().item() <= tol * scale


def test_param_spec_matches_reference_surface():
    sd = O.make_state_dict(0)
    assert len(sd) == 811                                                        # reference DKMv3.state_dict()
    assert sum(v.numel() for v in sd.values() if v.dtype.is_floating_point) == 70326800
    assert sd["decoder.conv_refiner.16.block1.0.weight"].shape == (1377, 1, 5, 5)  # 2*512 + 128 + 15^2, depthwise
    assert sd["decoder.conv_refiner.1.block1.0.weight"].shape == (24, 1, 5, 5)     # 12 -> 24, groups = 12


def test_stage_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, "dkm_stages.npz"))
    sd = O.make_state_dict(0)
    H, W = (int(v) for v in g["hw"])
    im0, im1 = O.seeded_pair(*(int(v) for v in g["image_hw"]), int(g["seed"]))
    with torch.no_grad():
        q, s = O._up(im0, (H, W)), O._up(im1, (H, W))
        pyr = O.resnet50_pyramid(sd, torch.cat((q, s)))
        _close(pyr[32], g["pyr32"])
        _close(pyr[2][:, ::8, ::4, ::4], g["pyr2_sub"])
        a = O._conv(sd, "decoder.proj.16", pyr[16])
        c = torch.cat((a.chunk(2)[1], a.chunk(2)[0]))
        _close(O.gp_forward(sd, "16", a, c), g["gp16"], 5e-4)
        flow0 = O.grid_coords(2, *a.shape[-2:]) + 0.05 * torch.randn(2, 2, *a.shape[-2:], generator=torch.Generator().manual_seed(1))
        _close(O.local_correlation(a, c, 7, flow0), g["local_corr"])
        cert, disp = O.conv_refiner(sd, "16", a, c, flow0)
        _close(cert, g["refiner_cert"], 2e-4)
        _close(disp, g["refiner_disp"], 2e-4)
    # local correlation, centre tap == plain dot product at the flow target (property)
    lc = torch.as_tensor(g["local_corr"])
    assert lc.shape[1] == 225


def test_match_and_sample_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, "dkm_match.npz"))
    sd = O.make_state_dict(0)
    H, W = (int(v) for v in g["hw"])
    up = tuple(int(v) for v in g["up"])
    im0, im1 = O.seeded_pair(*(int(v) for v in g["image_hw"]), int(g["seed"]))
    with torch.no_grad():
        cor = O.forward_symmetric(sd, O._up(im0, (H, W)), O._up(im1, (H, W)))
        _close(cor[16]["dense_flow"], g["flow16"], 5e-4)
        _close(cor[16]["dense_certainty"], g["cert16"], 5e-4)
        _close(cor[1]["dense_flow"], g["flow1"], 5e-4)
        warp, cert = O.match(sd, im0, im1, H, W, up)
    assert warp.shape == (up[0], 2 * up[1], 4) and cert.shape == (up[0], 2 * up[1])
    _close(warp[::2, ::2], g["warp"], 5e-4)
    _close(cert[::2, ::2], g["certainty"], 2e-3)
    # structure of match(): left half = (query grid, q->s flow), right half = (s->q flow, support grid)
    qc = O.grid_coords(1, *up).permute(0, 2, 3, 1)[0]
    assert torch.equal(warp[:, :up[1], :2], qc) and torch.equal(warp[:, up[1]:, 2:], qc)
    assert warp.abs().max() <= 1.0 and 0.0 <= cert.min() and cert.max() <= 1.0
    # sample(): same RNG stream as the reference run -> same draws (CPU generator), KDE pinned
    s = np.load(os.path.join(golden_dir, "dkm_sample.npz"))
    _close(O.kde(torch.as_tensor(s["matches"]), 0.1), s["kde"], 1e-5)
    out = O.gim_dkm_adapter(torch.as_tensor(s["matches"]), torch.as_tensor(s["certainty"]), (480, 640), (480, 640))
    assert out["mkpts0_f"].shape[1] == 2 and (out["mconf"] > 0).all()
    assert out["mkpts0_f"][:, 0].max() <= 640 and out["mkpts0_f"][:, 1].max() <= 480
