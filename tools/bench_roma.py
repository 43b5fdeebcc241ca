"""Secondary workload: gim_roma match() (+ sample) at the reference's configuration (RoMa(img_size=[672]), upsampling pass at
1344 x 1344), one GPU.
    python tools/bench_roma.py [--size 672] [--up 1344 1344 | --no-up] [--steps 3] [--precision bf16] [--pairs 1]
Prints one JSON line: pairs/s, ms per match(), stage split and the implicit-GEMM share (algorithmic FLOPs / its time)."""
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
    ap.add_argument("--size", type=int, default=672)
    ap.add_argument("--up", type=int, nargs=2, default=[1344, 1344])
    ap.add_argument("--no-up", action="store_true")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--layers", type=int, default=0, help="print the N most expensive implicit-GEMM layer shapes (stderr)")
    ap.add_argument("--stages", action="store_true", help="per-stage wall times (synchronising: run separately from the headline)")
    a = ap.parse_args()
    from gim_amd import ops
    from gim_amd.roma import RoMa, random_dinov2_weights
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = RoMa([a.size], precision=a.precision, dinov2_weights=random_dinov2_weights(dev)).eval()
    m.upsample_preds = not a.no_up
    with torch.no_grad():      # refiner outputs scaled down so the flow stays in range (what trained weights do)
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
    out = {"metric": "image-pairs/sec (gim_roma match + sample)", "value": a.pairs / (t_match + t_sample), "pairs_per_call": a.pairs,
           "unit": "pairs/s", "match_ms": t_match * 1e3, "sample_ms": t_sample * 1e3, "resolution": [a.size, a.size],
           "upsample_res": None if a.no_up else list(a.up), "precision": a.precision, "mean_certainty": float(cert.mean()),
           "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "data": "synthetic"}
    # implicit-GEMM share: algorithmic FLOPs and event time of every gim_conv2d_bn_act launch of one match()
    ops.PROFILE = []
    m.match_batch(im0, im1)
    torch.cuda.synchronize()
    prof = ops.PROFILE
    fl = sum(p[2] for p in prof)
    ms = sum(p[0].elapsed_time(p[1]) for p in prof)
    prof, ops.PROFILE = ops.PROFILE, None
    out.update({"igemm_tflop_per_match": fl / 1e12, "igemm_ms": ms, "igemm_tflops": fl / ms / 1e9})
    if a.layers:
        agg = {}
        for e0, e1, f, label in prof:
            t = e0.elapsed_time(e1)
            g = agg.setdefault(label, [0, 0.0, 0.0])
            g[0] += 1; g[1] += t; g[2] += f
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.layers]
        for label, (cnt, t, f) in rows:
            print(f"{label:40s} x{cnt:4d} {t:8.3f} ms {f / t / 1e9:8.1f} TFLOP/s", file=sys.stderr)
    if a.stages:
        st = {}

        def wrap(name, label):
            f = getattr(m, name)

            def g_(*args, **kw):
                torch.cuda.synchronize()
                t = time.perf_counter()
                r = f(*args, **kw)
                torch.cuda.synchronize()
                st[label] = st.get(label, 0.0) + (time.perf_counter() - t) * 1e3
                return r
            setattr(m, name, g_)
        for name, label in (("_vgg", "vgg_ms"), ("_dino_features", "dino_ms"), ("_coarse", "gp_decoder_ms"), ("_refine", "refiners_ms")):
            wrap(name, label)
        torch.cuda.synchronize()
        t = time.perf_counter()
        m.match_batch(im0, im1)
        torch.cuda.synchronize()
        st["total_sync_ms"] = (time.perf_counter() - t) * 1e3
        out["stages"] = {k: round(v, 2) for k, v in st.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
