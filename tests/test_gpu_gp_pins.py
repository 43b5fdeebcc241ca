"""Pins for the two dense matchers' only loose link, the Gaussian-process posterior.

gim_dkm / gim_roma match() agree with the reference's golden run to 2e-3 (warp) / 5e-3 (certainty), not to the 1e-4 that
every other stage reaches.  The gap is the GP posterior mu = K_xy (K_yy + 0.1 I)^-1 f and nothing else:

  1. `*_decoder_downstream_of_oracle_gp`: with the ORACLE's posterior injected in place of the engine's (test-side
     replacement of the module's `_gp`), everything downstream -- DFN / transformer decoder, the refiner cascade, local
     correlation, flow and certainty at every scale -- matches the oracle to 1e-4 of scale (measured 1e-6 ... 5e-6);
  2. `*_gp_posterior_vs_fp64`: on the SAME fp32 features, the engine's mu (fp32 cosine kernel on the MFMA, fp64 Cholesky) and
     the reference's arithmetic (fp32 cosine kernel, fp32 LU inverse, dkm.py:362 / roma.py:130) are each compared with an
     all-fp64 evaluation.  Measured: engine 6e-5 ... 2e-4, reference 4e-5 ... 8e-5 of scale.  K_yy + 0.1 I has a condition
     number of ~2e4, so fp32 rounding of the KERNEL MATRIX entries alone (1e-7 relative, whatever the summation order) moves mu
     by ~1e-4: two fp32 evaluations cannot reproduce each other to 1e-4 here, the reference itself is that far from exact
     arithmetic, and the decoder amplifies a 2e-4 difference in mu to the 2e-3 seen on the final warp.  (An earlier DESIGN
     claimed the reference's inverse was the noisier side; the measurement does not support that -- both sit at the fp32
     conditioning floor, the engine slightly above the reference.)
"""
import math
import os

import numpy as np
import pytest
import torch

import dkm_oracle as DO
import roma_oracle as RO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()


def _fp64_posterior(xr, yr, fr, sigma=0.1, T=0.2, eps=1e-6):
    """mu = K_xy (K_yy + sigma I)^-1 f in fp64 from fp32 feature rows [n, d] (cos kernel of dkm.py:135-144)"""
    x, y, f = xr.double(), yr.double(), fr.double()
    def k(a, b):
        c = a @ b.T / (a.norm(dim=-1)[:, None] * b.norm(dim=-1)[None] + eps)
        return ((c - 1.0) / T).exp()
    Kyy = k(y, y) + sigma * torch.eye(y.shape[0], dtype=torch.float64)
    return k(x, y) @ torch.linalg.solve(Kyy, f)


# ------------------------------------------------------------------------------------------------------ gim_dkm
def _dkm_setup():
    from gim_amd.dkm import DKMv3
    sd = DO.make_state_dict(0)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dkm_match.npz"))
    H, W = (int(v) for v in g["hw"])
    im0, im1 = DO.seeded_pair(*(int(v) for v in g["image_hw"]), int(g["seed"]))
    m = DKMv3(None, H, W, upsample_preds=False, precision="fp32")
    m.load_state_dict(sd)
    return m.eval(), sd, H, W, im0, im1


def _dkm_oracle_stage(sd, H, W, im0, im1):
    up = lambda t: torch.nn.functional.interpolate(t, size=(H, W), mode="bilinear", align_corners=False)  # noqa: E731
    with torch.no_grad():
        pyr = DO.resnet50_pyramid(sd, torch.cat((up(im0), up(im1))))
        swapped = {s: torch.cat((f.chunk(2)[1], f.chunk(2)[0])) for s, f in pyr.items()}
        proj = {s: (DO._conv(sd, f"decoder.proj.{s}", pyr[int(s)]), DO._conv(sd, f"decoder.proj.{s}", swapped[int(s)])) for s in ("32", "16")}
        mu = {s: DO.gp_forward(sd, s, *proj[s]) for s in ("32", "16")}
        cor = DO.decoder(sd, pyr, swapped)
    return proj, mu, cor


@pytest.mark.parametrize("exact", [True, False], ids=["fp64-gp", "fp32-entries"])
def test_dkm_gp_posterior_vs_fp64(exact):
    """exact: the parity mode's GP (entries, Cholesky and products in fp64, gim_gp_posterior_f64) -- at the formula's value;
    fp32-entries: the throughput path's arithmetic (fp32 kernel matrix on the MFMA, fp64 Cholesky) -- at the conditioning floor"""
    m, sd, H, W, im0, im1 = _dkm_setup()
    m.gp_exact = exact
    proj, mu_ref, _ = _dkm_oracle_stage(sd, H, W, im0, im1)
    m.match(im0.to(DEV), im1.to(DEV))            # packs the weights
    P, dt, _ = m._packed
    for s in ("32", "16"):
        a, c = proj[s]
        nb, _, h, w = a.shape
        n = h * w
        rows = a.permute(0, 2, 3, 1).reshape(nb * n, 512).contiguous()
        a32 = torch.zeros(nb * n + 64, 512, device=DEV)
        a32[:nb * n] = rows.to(DEV)
        out = torch.empty(nb * n, 256, device=DEV)
        m._gp(P, s, a32, nb, h, w, torch.float32, out)
        f = torch.cos(8 * math.pi * DO._conv(sd, f"decoder.gps.{s}.pos_conv", DO.grid_coords(1, h, w)))[0].flatten(1).T
        e_eng, e_ref = 0.0, 0.0
        for b in range(nb):
            mu64 = _fp64_posterior(rows[b * n:(b + 1) * n], c[b].flatten(1).T, f)
            scale = mu64.abs().max().item()
            e_eng = max(e_eng, (out[b * n:(b + 1) * n].cpu().double() - mu64).abs().max().item() / scale)
            e_ref = max(e_ref, (mu_ref[s][b].flatten(1).T.double() - mu64).abs().max().item() / scale)
        print(f"gim_dkm GP scale {s}: engine vs fp64 {e_eng:.2e}, reference arithmetic (fp32 inverse) vs fp64 {e_ref:.2e}")
        assert e_eng < (2e-6 if exact else 5e-4) and e_ref < 5e-4, (s, e_eng, e_ref)   # fp32 arithmetic: the conditioning floor


def test_dkm_decoder_downstream_of_oracle_gp():
    m, sd, H, W, im0, im1 = _dkm_setup()
    _, mu, cor = _dkm_oracle_stage(sd, H, W, im0, im1)

    def gp_from_oracle(P, s, a32, nb, h, w, tdt, out):
        out.copy_(mu[s].permute(0, 2, 3, 1).reshape(nb * h * w, 256).to(out.device))

    m._gp = gp_from_oracle                       # test-side replacement of ONE stage; the product module is untouched
    m.match(im0.to(DEV), im1.to(DEV))
    got = m._debug["corresps"]
    for s in (32, 16, 8, 4, 2, 1):
        ef = _rel(got[s][0].permute(0, 3, 1, 2), cor[s]["dense_flow"])
        ec = _rel(got[s][1].permute(0, 3, 1, 2), cor[s]["dense_certainty"])
        print(f"gim_dkm downstream of the oracle GP, scale {s}: flow {ef:.2e} certainty {ec:.2e}")
        assert ef < 1e-4 and ec < 1e-4, (s, ef, ec)   # measured 1e-6 ... 4e-6


# ------------------------------------------------------------------------------------------------------ gim_roma
def _roma_setup():
    from gim_amd.roma import RoMa
    sd, dsd = RO.make_state_dicts(0)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "roma_match.npz"))
    H, W = (int(v) for v in g["hw"])
    im0, im1 = DO.seeded_pair(*(int(v) for v in g["image_hw"]), int(g["seed"]))
    m = RoMa([H, W], precision="fp32", dinov2_weights=dsd)
    m.load_state_dict(sd)
    m.h_resized, m.w_resized, m.upsample_preds = H, W, False
    return m.eval(), sd, dsd, H, W, im0, im1


def _roma_oracle_stage(sd, dsd, H, W, im0, im1):
    up = lambda t: torch.nn.functional.interpolate(t, size=(H, W), mode="bilinear", align_corners=False)  # noqa: E731
    with torch.no_grad():
        pyr = RO.encoder(sd, dsd, torch.cat((up(im0), up(im1))))
        swapped = {s: torch.cat((f.chunk(2)[1], f.chunk(2)[0])) for s, f in pyr.items()}
        a, c = RO._proj(sd, "16", pyr[16]), RO._proj(sd, "16", swapped[16])
        mu = RO.gp_forward(sd, a, c)
        cor = RO.decoder(sd, pyr, swapped)
    return a, c, mu, cor


@pytest.mark.parametrize("exact", [True, False], ids=["fp64-gp", "fp32-entries"])
def test_roma_gp_posterior_vs_fp64(exact):
    m, sd, dsd, H, W, im0, im1 = _roma_setup()
    m.gp_exact = exact
    a, c, mu_ref, _ = _roma_oracle_stage(sd, dsd, H, W, im0, im1)
    nb, _, h, w = a.shape
    n = h * w
    rows = a.permute(0, 2, 3, 1).reshape(nb * n, 512).contiguous()
    a32 = torch.zeros(nb * n + 64, 512, device=DEV)
    a32[:nb * n] = rows.to(DEV)
    out = torch.empty(nb * n, 512, device=DEV)
    m.to(DEV)
    m._gp(a32, nb, h, w, out)
    f = torch.cos(8 * math.pi * RO._conv(sd, "decoder.gps.16.pos_conv", RO.grid_coords(1, h, w)))[0].flatten(1).T
    e_eng, e_ref = 0.0, 0.0
    for b in range(nb):
        mu64 = _fp64_posterior(rows[b * n:(b + 1) * n], c[b].flatten(1).T, f)
        scale = mu64.abs().max().item()
        e_eng = max(e_eng, (out[b * n:(b + 1) * n].cpu().double() - mu64).abs().max().item() / scale)
        e_ref = max(e_ref, (mu_ref[b].flatten(1).T.double() - mu64).abs().max().item() / scale)
    print(f"gim_roma GP: engine vs fp64 {e_eng:.2e}, reference arithmetic (fp32 inverse) vs fp64 {e_ref:.2e}")
    assert e_eng < (2e-6 if exact else 5e-4) and e_ref < 5e-4, (e_eng, e_ref)


def test_roma_decoder_downstream_of_oracle_gp():
    m, sd, dsd, H, W, im0, im1 = _roma_setup()
    _, _, mu, cor = _roma_oracle_stage(sd, dsd, H, W, im0, im1)

    def gp_from_oracle(a32, nb, h, w, out):
        out.copy_(mu.permute(0, 2, 3, 1).reshape(nb * h * w, 512).to(out.device))

    m._gp = gp_from_oracle
    m.match(im0.to(DEV), im1.to(DEV))
    low = m._debug["low"]
    gm_flow, gm_cert = low["gm"]
    assert _rel(gm_cert.permute(0, 3, 1, 2), cor[16]["gm_certainty"]) < 1e-4     # 5 transformer blocks on 1024-d tokens
    for s in (16, 8, 4, 2, 1):
        ef = _rel(low[s][0].permute(0, 3, 1, 2), cor[s]["flow"])
        ec = _rel(low[s][1].permute(0, 3, 1, 2), cor[s]["certainty"])
        print(f"gim_roma downstream of the oracle GP, scale {s}: flow {ef:.2e} certainty {ec:.2e}")
        assert ef < 1e-4 and ec < 1e-4, (s, ef, ec)   # measured 1e-7 ... 5e-6
