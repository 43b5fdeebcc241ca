"""Secondary workload: gim_dkm match() (+ sample) at the reference's configuration, one GPU.
    python tools/bench_dkm.py [--h 672 --w 896] [--up 1152 1536 | --no-up] [--steps 5] [--precision bf16]
Prints one JSON line: pairs/s, ms per match(), stage split (low-res pass / upsampling pass / sample) and error vs fp32."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=672)
    ap.add_argument("--w", type=int, default=896)
    ap.add_argument("--up", type=int, nargs=2, default=[1152, 1536])
    ap.add_argument("--no-up", action="store_true")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--pairs", type=int, default=1)
    a = ap.parse_args()
    from gim_amd.dkm import DKMv3
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = DKMv3(None, a.h, a.w, upsample_preds=not a.no_up, precision=a.precision).eval()
    # random init, refiner outputs scaled down so the flow stays in range (what trained weights do)
    with torch.no_grad():
        for s in ("16", "8", "4", "2", "1"):
            m.decoder.conv_refiner[s].out_conv.weight.mul_(0.05)
            m.decoder.conv_refiner[s].out_conv.bias.mul_(0.05)
    if not a.no_up:
        m.upsample_res = tuple(a.up)
    g = torch.Generator().manual_seed(1)
    base = torch.nn.functional.interpolate(torch.rand(1, 3, 60, 80, generator=g), size=(480, 640), mode="bicubic").clamp(0.05, 1)
    im0 = base.to(dev).expand(a.pairs, -1, -1, -1).contiguous()
    im1 = torch.roll(base, shifts=(12, 20), dims=(2, 3)).to(dev).expand(a.pairs, -1, -1, -1).contiguous()
    for _ in range(2):
        warp, cert = m.match_batch(im0, im1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        warp, cert = m.match_batch(im0, im1)
    torch.cuda.synchronize()
    t_match = (time.perf_counter() - t0) / a.steps
    m.sample(warp[0], cert[0], 5000)      # warm-up (first-use module loads of the sort / index kernels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        for b in range(a.pairs):
            sm, sc = m.sample(warp[b], cert[b], 5000)
    torch.cuda.synchronize()
    t_sample = (time.perf_counter() - t0) / a.steps
    flops = 5.3e12 if not a.no_up else 1.56e12   # 1.56 TFLOP low-res + 3.71 TFLOP upsampling pass
    print(json.dumps({"metric": "image-pairs/sec (gim_dkm match + sample)", "value": a.pairs / (t_match + t_sample), "pairs_per_call": a.pairs, "unit": "pairs/s",
                      "match_ms": t_match * 1e3, "sample_ms": t_sample * 1e3, "resolution": [a.h, a.w],
                      "upsample_res": None if a.no_up else list(a.up), "precision": a.precision,
                      "achieved_tflops": a.pairs * flops / t_match / 1e12, "mean_certainty": float(cert.mean()), "data": "synthetic"}))


if __name__ == "__main__":
    main()
