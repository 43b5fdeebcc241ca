"""TEST INFRASTRUCTURE ONLY -- pins `oracle/dkm_oracle.py` to the reference's own DKMv3.

Run by hand in the authoring container (needs /root/reference):  python oracle/make_golden_dkm.py
Builds the reference model through `oracle/ref_shims.py::reference_dkm`, loads `make_state_dict(0)` with
strict=True, runs reference and restatement stage by stage and end to end on seeded inputs, asserts agreement and
stores the REFERENCE's outputs in tests/golden/dkm_*.npz.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dkm_oracle as O  # noqa: E402
from oracle import ref_shims  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def close(a, b, tol, what):
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, (what, err, scale)
    return err


@torch.no_grad()
def main():
    H, W, UP = 128, 160, (192, 256)
    ref = ref_shims.reference_dkm(H, W, upsample_res=UP)
    sd = O.make_state_dict(0)
    spec = O.dkm_param_spec()
    rsd = ref.state_dict()
    assert set(rsd) == set(spec) and all(tuple(rsd[k].shape) == tuple(spec[k]) for k in spec), "parameter surface differs"
    ref.load_state_dict(sd, strict=True)

    im0, im1 = O.seeded_pair(160, 224, 3)
    # ---- stages -------------------------------------------------------------------------------------------
    q = torch.nn.functional.interpolate(im0, size=(H, W), mode="bilinear", align_corners=False)
    s = torch.nn.functional.interpolate(im1, size=(H, W), mode="bilinear", align_corners=False)
    pyr_ref = ref.encoder(torch.cat((q, s)))
    pyr = O.resnet50_pyramid(sd, torch.cat((q, s)))
    for k in pyr_ref:
        close(pyr[k], pyr_ref[k], 1e-5, f"pyramid {k}")
    a = ref.decoder.proj["16"](pyr_ref[16]); c = torch.cat((a.chunk(2)[1], a.chunk(2)[0]))
    gp_ref = ref.decoder.gps["16"](a, c)
    close(O.gp_forward(sd, "16", a, c), gp_ref, 2e-4, "gp16")
    flow0 = O.grid_coords(2, *a.shape[-2:]) + 0.05 * torch.randn(2, 2, *a.shape[-2:], generator=torch.Generator().manual_seed(1))
    from networks.dkm.utils.local_correlation import local_correlation as ref_lc
    lc_ref = ref_lc(a, c, local_radius=7, flow=flow0)
    close(O.local_correlation(a, c, 7, flow0), lc_ref, 1e-5, "local_corr")
    rc_ref, rd_ref = ref.decoder.conv_refiner["16"](a, c, flow0)
    oc, od = O.conv_refiner(sd, "16", a, c, flow0)
    close(oc, rc_ref, 1e-4, "refiner cert"); close(od, rd_ref, 1e-4, "refiner disp")
    np.savez_compressed(os.path.join(OUT, "dkm_stages.npz"), seed=3, hw=np.array([H, W]), image_hw=np.array([160, 224]),
                        pyr32=pyr_ref[32].numpy(), pyr2_sub=pyr_ref[2][:, ::8, ::4, ::4].numpy(),
                        gp16=gp_ref.numpy(), local_corr=lc_ref.numpy(), refiner_cert=rc_ref.numpy(), refiner_disp=rd_ref.numpy())

    # ---- decoder + match ----------------------------------------------------------------------------------------
    cor_ref = ref.forward_symmetric({"query": q, "support": s}, batched=True)
    cor = O.forward_symmetric(sd, q, s)
    e_f = max(close(cor[k]["dense_flow"], cor_ref[k]["dense_flow"], 2e-4, f"flow {k}") for k in cor_ref)
    e_c = max(close(cor[k]["dense_certainty"], cor_ref[k]["dense_certainty"], 2e-4, f"cert {k}") for k in cor_ref)
    warp_ref, cert_ref = ref.match(im0, im1)
    warp, cert = O.match(sd, im0, im1, H, W, UP)
    e_w, e_p = close(warp, warp_ref, 2e-4, "warp"), close(cert, cert_ref, 5e-4, "certainty")
    print(f"decoder flow {e_f:.2e} cert {e_c:.2e}; match warp {e_w:.2e} certainty {e_p:.2e}; "
          f"mean certainty {cert_ref.mean():.3f}, in-range flow {(warp_ref[..., 2:].abs() < 1).float().mean():.3f}")
    np.savez_compressed(os.path.join(OUT, "dkm_match.npz"), seed=3, hw=np.array([H, W]), up=np.array(UP),
                        image_hw=np.array([160, 224]), flow16=cor_ref[16]["dense_flow"].numpy(),
                        cert16=cor_ref[16]["dense_certainty"].numpy(), flow1=cor_ref[1]["dense_flow"].numpy(),
                        warp=warp_ref[::2, ::2].numpy().astype(np.float32), certainty=cert_ref[::2, ::2].numpy())

    # ---- sample(): same global RNG stream, CPU --------------------------------------------------------------------
    torch.manual_seed(123)
    sm_ref, sc_ref = ref.sample(warp_ref, cert_ref, 500)
    torch.manual_seed(123)
    sm, sc = O.sample(warp_ref, cert_ref, 500)
    assert torch.equal(sm, sm_ref) and torch.equal(sc, sc_ref)
    dens = O.kde(sm_ref, 0.1)
    np.savez_compressed(os.path.join(OUT, "dkm_sample.npz"), matches=sm_ref.numpy(), certainty=sc_ref.numpy(), kde=dens.numpy())
    print("sample: exact under the same RNG stream;", sm_ref.shape)


if __name__ == "__main__":
    main()
