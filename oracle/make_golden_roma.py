"""TEST INFRASTRUCTURE ONLY -- pins `oracle/roma_oracle.py` to the reference's own RoMa.

Run by hand in the authoring container (needs /root/reference):  python oracle/make_golden_roma.py
Builds the reference model through `oracle/ref_shims.py::reference_roma` (the DINOv2 weights it would download are the
seeded `dino_sd`), loads `make_state_dicts(0)[0]` with strict=True, runs reference and restatement on seeded inputs,
asserts agreement and stores the REFERENCE's outputs in tests/golden/roma_*.npz.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import roma_oracle as O  # noqa: E402
import dkm_oracle as DO  # noqa: E402
import ref_shims  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def close(a, b, tol, what):
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, (what, err, scale)
    return err


@torch.no_grad()
def main():
    H, W, UP = 112, 140, (168, 224)          # multiples of 14 (DINOv2 patches) and of 8 (VGG pyramid)
    sd, dsd = O.make_state_dicts(0)
    ref = ref_shims.reference_roma(H, W, dsd, upsample_res=UP)
    rsd = ref.state_dict()
    spec = O.roma_param_spec()
    assert set(rsd) == set(spec), (sorted(set(rsd) ^ set(spec))[:10])
    assert all(tuple(rsd[k].shape) == tuple(spec[k]) for k in spec)
    ref.load_state_dict(sd, strict=True)
    dino = ref.encoder.dinov2_vitl14[0]
    assert {k: tuple(v.shape) for k, v in dino.state_dict().items()} == {k: tuple(v) for k, v in O.dino_param_spec().items()}

    im0, im1 = DO.seeded_pair(150, 200, 4)
    up = lambda t, s: torch.nn.functional.interpolate(t, size=s, mode="bilinear", align_corners=False)  # noqa: E731
    q, s_ = up(im0, (H, W)), up(im1, (H, W))
    X = torch.cat((q, s_))
    # ---- encoders ---------------------------------------------------------------------------------------------
    pyr_ref = ref.encoder(X)
    pyr = O.encoder(sd, dsd, X)
    for k in pyr_ref:
        close(pyr[k], pyr_ref[k], 2e-5, f"pyramid {k}")
    # ---- decoder pieces -----------------------------------------------------------------------------------------
    a = ref.decoder.proj["16"](pyr_ref[16]); c = torch.cat((a.chunk(2)[1], a.chunk(2)[0]))
    gp_ref = ref.decoder.gps["16"](a, c)
    close(O.gp_forward(sd, a, c), gp_ref, 2e-4, "gp")
    cls_ref, cert_ref, _ = ref.decoder.embedding_decoder(gp_ref, a, None, "16")
    cls, cert = O.transformer_decoder(sd, gp_ref, a)
    close(cls, cls_ref, 2e-4, "cls"); close(cert, cert_ref, 2e-4, "gm certainty")
    from networks.roma.roma import cls_to_flow_refine as ref_c2f
    flow_ref = ref_c2f(cls_ref)
    close(O.cls_to_flow_refine(cls_ref), flow_ref, 1e-5, "cls_to_flow")
    np.savez_compressed(os.path.join(OUT, "roma_stages.npz"), seed=4, hw=np.array([H, W]), image_hw=np.array([150, 200]),
                        dino16=pyr_ref[16].numpy(), vgg8_sub=pyr_ref[8][:, ::8].numpy(), vgg1_sub=pyr_ref[1][:, ::16, ::4, ::4].numpy(),
                        gp=gp_ref.numpy(), cls_sub=cls_ref[:, ::64].numpy(), gm_certainty=cert_ref.numpy(), gm_flow=flow_ref.numpy())
    # ---- full match --------------------------------------------------------------------------------------------------
    cor_ref = ref.forward_symmetric({"im_A": q, "im_B": s_})
    cor = O.forward_symmetric(sd, dsd, q, s_)
    e1 = max(close(cor[k]["flow"], cor_ref[k]["flow"], 5e-4, f"flow {k}") for k in cor_ref)
    e2 = max(close(cor[k]["certainty"], cor_ref[k]["certainty"], 5e-4, f"cert {k}") for k in cor_ref)
    warp_ref, c_ref = ref.match(im0, im1)
    warp, cc = O.match(sd, dsd, im0, im1, H, W, UP)
    e3, e4 = close(warp, warp_ref, 5e-4, "warp"), close(cc, c_ref, 1e-3, "certainty")
    print(f"decoder flow {e1:.2e} cert {e2:.2e}; match warp {e3:.2e} certainty {e4:.2e}; mean certainty {c_ref.mean():.3f}, "
          f"in-range {(warp_ref[..., 2:].abs() < 1).float().mean():.3f}")
    np.savez_compressed(os.path.join(OUT, "roma_match.npz"), seed=4, hw=np.array([H, W]), up=np.array(UP), image_hw=np.array([150, 200]),
                        flow16=cor_ref[16]["flow"].numpy(), cert16=cor_ref[16]["certainty"].numpy(), flow1=cor_ref[1]["flow"].numpy(),
                        warp=warp_ref[::2, ::2].numpy(), certainty=c_ref[::2, ::2].numpy())


if __name__ == "__main__":
    main()
