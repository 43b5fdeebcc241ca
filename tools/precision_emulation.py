"""Where do the bf16-mode index flips of gim_loftr come from?  CPU study, no GPU needed.

Runs the fp32 CPU oracle (test infrastructure) on one match-rich 640x480 pair, then re-runs it with the operands of
selected stages rounded the way the engine's 16-bit modes round them (BatchNorm folded into the conv weights first,
weights and every conv / linear input rounded, ReLU outputs = stored activations rounded, attention operands
rounded) and prints the match-set flip rate of every variant against the unrounded run.

Formats: 'bf16' (8 significand bits), 'fp16' (11 bits), 'bf16x2' (hi + lo bf16 pair = 16 bits: what a 3-product
split bf16 MFMA carries), 'fp32' (no rounding).  Stages: stem, layer1, layer2, layer3, fpn, transformer, sim (profiles/r03_precision_emulation_stage1.txt: the first run, per-stage
isolation with stem and layer1 as one stage).

    python tools/precision_emulation.py [n_pairs]   ->  table on stdout (profiles/r03_precision_emulation.txt)
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import loftr_oracle as O  # noqa: E402
from tools import synth_loftr as S  # noqa: E402
from tools.parity import parity_vs_oracle  # noqa: E402


def rnd(x, fmt):
    if fmt == "fp32":
        return x
    if fmt == "bf16":
        return x.bfloat16().float()
    if fmt == "fp16":
        return x.half().float()
    if fmt == "bf16x2":
        hi = x.bfloat16().float()
        return hi + (x - hi).bfloat16().float()
    if fmt == "fp16x2":      # hi + lo IEEE-fp16 pair (22 significand bits): what the split-operand MFMA of precision='mixed' carries
        hi = x.half().float()
        return hi + (x - hi).half().float()
    raise ValueError(fmt)


def fold_bn(sd):
    """conv weight *= gamma / sqrt(var + eps); the BatchNorm that follows becomes a bias add (same fp32 function up to
    rounding) -- the engine rounds the FOLDED weights (gim_amd/packing.py)."""
    sd = {k: v.clone() for k, v in sd.items()}
    pairs = []
    for k in sd:
        if k.endswith(".running_mean"):
            bn = k[:-len(".running_mean")]
            if bn.endswith(".bn1") and ".layer" not in bn:
                conv = bn[:-3] + "conv1"
            elif bn[-4:-1] == ".bn":
                conv = bn[:-4] + ".conv" + bn[-1]
            elif bn.endswith("downsample.1"):
                conv = bn[:-1] + "0"
            elif bn.endswith("outconv2.1"):
                conv = bn[:-1] + "0"
            else:
                raise KeyError(bn)
            pairs.append((conv, bn))
    for conv, bn in pairs:
        s = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + 1e-5)
        sd[conv + ".weight"] = sd[conv + ".weight"] * s[:, None, None, None]
        sd[bn + ".running_mean"] = sd[bn + ".running_mean"] * s
        sd[bn + ".running_var"] = torch.full_like(s, 1.0 - 1e-5)
        sd[bn + ".weight"] = torch.ones_like(s)
    return sd


def parts(f):
    """'fp16' -> the same format for weights, conv / linear inputs and stored ReLU outputs; 'w/x/s' (e.g. 'fp16x2/fp16/fp16'):
    one format each -- split WEIGHTS cost MFMA work only, split activations cost bytes too"""
    p = f.split("/")
    return (p[0], p[0], p[0]) if len(p) == 1 else (p[0], p[1], p[2])


class Emu:
    """Patches the oracle module's F.conv2d / F.linear / F.relu / linear_attention / similarity by stage."""

    def __init__(self, sd, fmts):
        self.fmts = fmts      # stage -> format
        self.sd = sd
        self.stage_of = {}    # id(weight tensor) -> stage
        for k, v in sd.items():
            if not k.endswith("weight") or v.dim() < 2:
                continue
            # hierarchical stage names: 'layer2.1' (Bottleneck), 'fpn.layer2_outconv2.0', 'transformer.5'; a format table may
            # name the full stage or only its group ('layer2', 'fpn', 'transformer')
            if k.startswith("backbone.encode.layer"):
                st = ".".join(k.split(".")[2:4])
            elif k.startswith("backbone.encode"):
                st = "stem"
            elif k.startswith("backbone"):
                st = "fpn." + k[len("backbone."):-len(".weight")]
            elif k.startswith("loftr_coarse"):
                st = "transformer." + k.split(".")[2]
            else:
                st = "fine"
            self.stage_of[id(v)] = st
        self.cur = "stem"
        self.keep_stream = fmts.get("keep_stream", False)   # True: the ResNet residual stream x (ReLU outputs) is NOT rounded

    def look(self, st):
        if st in self.fmts:
            return self.fmts[st]
        return self.fmts.get(st.split(".")[0], "fp32")

    def fmt(self, w):
        st = self.stage_of.get(id(w), "fine")
        self.cur = st
        return self.look(st)

    def run(self, data):
        conv0, lin0, relu0, la0, cm0 = F.conv2d, F.linear, F.relu, O.linear_attention, O.conf_matrix_dual_softmax
        emu = self

        class FF:
            def __getattr__(self, n):
                return getattr(F, n)

            @staticmethod
            def conv2d(x, w, *a, **k):
                wf, xf, _ = parts(emu.fmt(w))
                return conv0(rnd(x, xf), rnd(w, wf), *a, **k)

            @staticmethod
            def linear(x, w, *a, **k):
                wf, xf, _ = parts(emu.fmt(w))
                return lin0(rnd(x, xf), rnd(w, wf), *a, **k)

            @staticmethod
            def relu(x, *a, **k):
                if emu.keep_stream and emu.cur.split(".")[0] in ("stem", "layer1", "layer2", "layer3"):
                    return relu0(x)
                return rnd(relu0(x), parts(emu.look(emu.cur))[2])

        def la(q, k, v, q_mask=None, kv_mask=None, eps=1e-6):
            f = parts(emu.look(emu.cur))[1]   # 'transformer.N' or 'fine'
            Q = rnd(F.elu(q) + 1, f)
            K = rnd(F.elu(k) + 1, f)
            v = rnd(v, f)
            vl = v.size(1)
            KV = torch.einsum("nshd,nshv->nhdv", K, v / vl)
            Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + eps)
            return (torch.einsum("nlhd,nhdv,nlh->nlhv", Q, rnd(KV, f), Z) * vl).contiguous()

        def cm(f0, f1, temperature=0.1, m0=None, m1=None):
            f = parts(emu.fmts.get("sim", "fp32"))[1]
            return cm0(rnd(f0, f), rnd(f1, f), temperature, m0, m1)

        O.F = FF()
        O.linear_attention = la
        O.conf_matrix_dual_softmax = cm
        try:
            with torch.no_grad():
                return O.loftr_forward(self.sd, data)
        finally:
            O.F = F
            O.linear_attention = la0
            O.conf_matrix_dual_softmax = cm0


def main():
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    torch.set_num_threads(os.cpu_count() or 1)
    model, sd = S.synthetic_model("fp32")
    sdf = fold_bn(sd)
    c0, c1 = S.textured_pairs(npairs, 480, 640, seed=1234, frac=0.45)

    def data():
        return {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}

    t = time.time()
    with torch.no_grad():
        ref = O.loftr_forward(sd, data())
    print(f"oracle: {ref['b_ids'].numel()} matches over {npairs} pair(s), {time.time() - t:.1f} s", flush=True)
    ST = ("stem", "layer1", "layer2", "layer3", "fpn", "transformer", "sim")
    BB = ST[:5]
    allf = lambda f: {s_: f for s_ in ST}   # noqa: E731
    variants = [("all bf16", allf("bf16")),
                ("all fp16", allf("fp16")),
                ("backbone fp16, transformer+sim bf16", {**{s_: "fp16" for s_ in BB}, "transformer": "bf16", "sim": "bf16"}),
                ("backbone+sim fp16, transformer bf16", {**{s_: "fp16" for s_ in BB}, "transformer": "bf16", "sim": "fp16"}),
                ("all fp16, sim fp32", {**allf("fp16"), "sim": "fp32"}),
                ("stem fp32, rest bf16", {**allf("bf16"), "stem": "fp32"}),
                ("stem fp16, rest bf16", {**allf("bf16"), "stem": "fp16"}),
                ("stem+layer1 fp16, rest bf16", {**allf("bf16"), "stem": "fp16", "layer1": "fp16"}),
                ("stem+layer1 fp32, rest bf16", {**allf("bf16"), "stem": "fp32", "layer1": "fp32"}),
                ("all bf16, unrounded ResNet residual stream", {**allf("bf16"), "keep_stream": True}),
                ("all bf16, unrounded stream, stem fp32", {**allf("bf16"), "stem": "fp32", "keep_stream": True}),
                ]
    if len(sys.argv) > 2 and sys.argv[2] == "sweep":
        # which layers carry the fp16 mode's flips?  (profiles/r04_precision_sweep.txt)
        base = sys.argv[3] if len(sys.argv) > 3 else "fp16"
        groups = ("stem", "layer1", "layer2", "layer3", "fpn", "transformer", "sim")
        variants = [(f"all {base}", allf(base))]
        variants += [(f"only {g} {base}, rest fp32", {g: base}) for g in groups]
        variants += [(f"all {base} except {g} fp32", {**allf(base), g: "fp32"}) for g in groups]
        blocks = ["layer1.0", "layer1.1", "layer1.2", "layer2.0", "layer2.1", "layer2.2", "layer2.3"] + [f"layer3.{i}" for i in range(6)] + \
                 ["fpn.layer3_outconv", "fpn.layer2_outconv", "fpn.layer2_outconv2.0", "fpn.layer2_outconv2.3"] + [f"transformer.{i}" for i in range(8)]
        variants += [(f"only {g} {base}, rest fp32", {g: base}) for g in blocks]
    if len(sys.argv) > 2 and sys.argv[2] == "mixed":
        # candidate 'mixed' modes: split (hi + lo fp16) operands on the sensitive stages
        x2 = lambda *gs: {**allf("fp16"), **{g: "fp16x2" for g in gs}}   # noqa: E731
        variants = [("all fp16", allf("fp16")),
                    ("all fp16x2", allf("fp16x2")),
                    ("stem fp16x2, rest fp16", x2("stem")),
                    ("stem+layer1 fp16x2, rest fp16", x2("stem", "layer1")),
                    ("stem+layer1+sim fp16x2, rest fp16", x2("stem", "layer1", "sim")),
                    ("stem+layer1+layer2 fp16x2, rest fp16", x2("stem", "layer1", "layer2")),
                    ("backbone fp16x2, transformer+sim fp16", x2("stem", "layer1", "layer2", "layer3", "fpn")),
                    ("backbone+sim fp16x2, transformer fp16", x2("stem", "layer1", "layer2", "layer3", "fpn", "sim")),
                    ("transformer+sim fp16x2, backbone fp16", x2("transformer", "sim")),
                    ]
    if len(sys.argv) > 2 and sys.argv[2] == "mixed2":
        # with the stem on split operands: what else is worth spending precision on?  w/x/s = weights / inputs / stored activations
        st = {**allf("fp16"), "stem": "fp16x2"}
        BBL = ("layer1", "layer2", "layer3")
        variants = [("stem x2, rest fp16", st),
                    ("stem x2 + layer1-3 split WEIGHTS", {**st, **{g: "fp16x2/fp16/fp16" for g in BBL}}),
                    ("stem x2 + layer1-3 split conv INPUTS", {**st, **{g: "fp16/fp16x2/fp16" for g in BBL}}),
                    ("stem x2 + layer1-3 unrounded residual STREAM", {**st, **{g: "fp16/fp16/fp32" for g in BBL}}),
                    ("stem x2 + layer1-3 weights + stream", {**st, **{g: "fp16x2/fp16/fp32" for g in BBL}}),
                    ("stem x2 + layer1 split weights", {**st, "layer1": "fp16x2/fp16/fp16"}),
                    ("stem x2 + all split WEIGHTS (backbone, transformer)", {**st, **{g: "fp16x2/fp16/fp16" for g in BBL + ("fpn", "transformer")}}),
                    ("stem x2 + sim x2", {**st, "sim": "fp16x2"}),
                    ("stem x2 + sim x2 + transformer x2", {**st, "sim": "fp16x2", "transformer": "fp16x2"}),
                    ]
    if len(sys.argv) > 2 and sys.argv[2] == "stem":
        # which half of the stem's split matters: the image (x) or the filters (w)?  (6 channels [x_hi | x_lo] x [w_hi | w_hi] fit the
        # 8-channel image layout of the plain stem; the full split needs 9 -> 16 channels and twice the K of the first convolution)
        f = allf("fp16")
        variants = [("stem plain fp16", f),
                    ("stem split IMAGE only (w fp16, x hi+lo)", {**f, "stem": "fp16/fp16x2/fp16"}),
                    ("stem split WEIGHTS only (w hi+lo, x fp16)", {**f, "stem": "fp16x2/fp16/fp16"}),
                    ("stem split both (stored output fp16)", {**f, "stem": "fp16x2/fp16x2/fp16"}),
                    ]
    for name, fm in variants:
        t = time.time()
        out = Emu(sdf, fm).run(data())
        fr, dc, mx, n = [], [], [], 0
        for b in range(npairs):
            p = parity_vs_oracle(out, ref, b, b)
            fr.append(p["flip_rate"])
            dc.append(p.get("mean_abs_dmconf", 0.0))
            mx.append(p.get("max_abs_dmconf", 0.0))
            n += p["engine_matches"]
        print(f"{name:58s} flip {100 * sum(fr) / len(fr):6.3f} %   mean|dmconf| {sum(dc) / len(dc):.5f}   max|dmconf| {max(mx):.4f}   matches {n}   "
              f"({time.time() - t:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
