"""LightGlue on MI355X: the reference's module surface (`networks/lightglue/models/matchers/lightglue.py:303-545`)
over hand-written HIP.

Drop-in contract:
  * `LightGlue(conf_dict)` with the reference's `default_conf`; `state_dict()` has the reference's 251 tensors
    (posenc.Wr, transformers.{i}.{self_attn,cross_attn}.*, log_assignment.{i}.*, token_confidence.{i}.*), so the
    `model.`-stripped halves of gim_lightglue checkpoints load unchanged (`demo.py:388-395`);
  * `model(data)` needs keypoints0/1 [B,M|N,2], descriptors0/1 [B,M|N,256] and image_size0/1 or resize0/1
    (flipped with `[:, [1, 0]]` exactly like lightglue.py:414-415) and returns the reference's dict: matches0/1
    (int64, -1 = unmatched), matching_scores0/1, matches / scores (per-pair lists), ref_descriptors0/1, stop,
    prune0/1 and `log_assignment` (materialised on demand, see `LazyLogAssignment`);
  * gim runs LightGlue with depth_confidence = width_confidence = -1 (no early stop / pruning,
    `demo.py:345-349`): only that configuration is built.

Per layer and image set the data path is: fused QKV GEMM -> rotary (in place) -> V transpose -> flash SDPA
(MFMA, scores never leave the CU) -> out-proj GEMM -> FFN GEMM -> LayerNorm+GELU -> FFN GEMM with the residual
added in its epilogue; the assignment head is `lg_assign.hip` (fused double log-softmax / mutual arg-max).
Both images' keypoints are stacked into one row block so every GEMM is a single launch.
No CPU / eager fallback.
"""
import os

import torch

from ..precision import resolve as resolve_precision
import torch.nn as nn

from .. import ops
from .._lib import ACT_GELU, ACT_NONE, GIM_BF16, GIM_F16, GIM_F32, GimHipError
from ..packing import pack_conv, torch_dtype


class _FFNBlock(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.ffn = nn.Sequential(nn.Linear(2 * d, 2 * d), nn.LayerNorm(2 * d, elementwise_affine=True), nn.GELU(),
                                 nn.Linear(2 * d, d))


class _SelfBlock(_FFNBlock):
    """parameter layout of lightglue.py:121-139"""

    def __init__(self, d):
        super().__init__(d)
        self.Wqkv = nn.Linear(d, 3 * d)
        self.out_proj = nn.Linear(d, d)


class _CrossBlock(_FFNBlock):
    """parameter layout of lightglue.py:159-181"""

    def __init__(self, d):
        super().__init__(d)
        self.to_qk = nn.Linear(d, d)
        self.to_v = nn.Linear(d, d)
        self.to_out = nn.Linear(d, d)


class _TransformerLayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.self_attn = _SelfBlock(d)
        self.cross_attn = _CrossBlock(d)


class _MatchAssignment(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.matchability = nn.Linear(d, 1)
        self.final_proj = nn.Linear(d, d)


class _TokenConfidence(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.token = nn.Sequential(nn.Linear(d, 1), nn.Sigmoid())


class _PosEnc(nn.Module):
    def __init__(self, m, f_dim):
        super().__init__()
        self.Wr = nn.Linear(m, f_dim // 2, bias=False)
        nn.init.normal_(self.Wr.weight.data, mean=0, std=1.0)


class LazyLogAssignment:
    """pred['log_assignment'] ([B, M+1, N+1] fp32, 16.8 MB per pair at 2048 keypoints): no caller of the gim
    pipelines reads it (`trainer/lightning.py:175-185`, `hloc/match_features.py:156-160`), so it is produced by
    `.get()` on demand from the cached softmax statistics' inputs."""

    def __init__(self, assign_result):
        self._r, self._t = assign_result, None

    def get(self):
        if self._t is None:
            self._t = ops.lg_log_assignment(self._r)
        return self._t

    def __array__(self, dtype=None):
        a = self.get().cpu().numpy()
        return a.astype(dtype) if dtype is not None else a


class LightGlue(nn.Module):
    default_conf = {
        "name": "lightglue", "input_dim": 256, "add_scale_ori": False, "descriptor_dim": 256, "n_layers": 9,
        "num_heads": 4, "flash": False, "mp": False, "depth_confidence": -1, "width_confidence": -1,
        "filter_threshold": 0.0, "checkpointed": False, "weights": "superpoint_lightglue",
        "weights_from_version": "v0.1_arxiv", "loss": {"gamma": 1.0, "fn": "nll", "nll_balancing": 0.5},
    }
    required_data_keys = ["keypoints0", "keypoints1", "descriptors0", "descriptors1"]

    def __init__(self, conf):
        super().__init__()
        self.conf = c = {**self.default_conf, **dict(conf)}
        if c["depth_confidence"] > 0 or c["width_confidence"] > 0:
            raise NotImplementedError("early stopping / point pruning are disabled in gim (demo.py:345-349) and not built")
        if c["add_scale_ori"] or c["input_dim"] != c["descriptor_dim"] or c["descriptor_dim"] != 256 or c["num_heads"] != 4:
            raise NotImplementedError("only the SuperPoint configuration (256-d, 4 heads, no scale/orientation) is built")
        self.precision = resolve_precision(c.get("precision"), "LightGlue")
        d, n = c["descriptor_dim"], c["n_layers"]
        self.input_proj = nn.Identity()
        self.posenc = _PosEnc(2, d // c["num_heads"])
        self.transformers = nn.ModuleList([_TransformerLayer(d) for _ in range(n)])
        self.log_assignment = nn.ModuleList([_MatchAssignment(d) for _ in range(n)])
        self.token_confidence = nn.ModuleList([_TokenConfidence(d) for _ in range(n - 1)])
        self._packed = None

    def load_state_dict(self, state_dict, *args, **kwargs):
        self._packed = None
        return super().load_state_dict(state_dict, *args, **kwargs)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    # ---- one-time weight packing ----------------------------------------------------------------------------
    def _prepack(self, device):
        dt = {"bf16": GIM_BF16, "fp16": GIM_F16, "fp32": GIM_F32}[self.precision]
        d, H = 256, 4
        dh = d // H
        layers = []

        def lin(w, b):
            return pack_conv(w.detach(), None, dt, device, bias=b.detach())

        # Wqkv output feature f = h*(3*dh) + j*3 + s (s = q,k,v; lightglue.py:147-148) -> [q | k | v], head-major
        perm = torch.empty(3 * d, dtype=torch.long)
        for s in range(3):
            for h in range(H):
                for j in range(dh):
                    perm[s * d + h * dh + j] = h * 3 * dh + j * 3 + s
        for tl in self.transformers:
            sa, ca = tl.self_attn, tl.cross_attn
            p = {
                "s_qkv": lin(sa.Wqkv.weight[perm], sa.Wqkv.bias[perm]),
                "s_out": lin(sa.out_proj.weight, sa.out_proj.bias),
                "s_f0": lin(sa.ffn[0].weight, sa.ffn[0].bias), "s_f3": lin(sa.ffn[3].weight, sa.ffn[3].bias),
                "s_ln": (sa.ffn[1].weight.detach().float().to(device).contiguous(), sa.ffn[1].bias.detach().float().to(device).contiguous()),
                "c_qkv": lin(torch.cat([ca.to_qk.weight, ca.to_v.weight], 0), torch.cat([ca.to_qk.bias, ca.to_v.bias], 0)),
                "c_out": lin(ca.to_out.weight, ca.to_out.bias),
                "c_f0": lin(ca.ffn[0].weight, ca.ffn[0].bias), "c_f3": lin(ca.ffn[3].weight, ca.ffn[3].bias),
                "c_ln": (ca.ffn[1].weight.detach().float().to(device).contiguous(), ca.ffn[1].bias.detach().float().to(device).contiguous()),
            }
            layers.append(p)
        la = self.log_assignment[self.conf["n_layers"] - 1]
        head = {"final": lin(la.final_proj.weight, la.final_proj.bias),
                "mw": la.matchability.weight.detach().float().reshape(-1).to(device).contiguous(),
                "mb": la.matchability.bias.detach().float().reshape(-1).to(device).contiguous()}
        wr = self.posenc.Wr.weight.detach().float().to(device).contiguous()
        self._packed = (layers, head, wr, dt, device)

    def _ffn(self, p, pre, CAT, X32, HID, HID2, alias):
        ops.linear(CAT, p[pre + "_f0"], HID)
        ops.layernorm_act(HID, p[pre + "_ln"][0], p[pre + "_ln"][1], HID2, ACT_GELU)
        R = X32.shape[0]
        ops.conv_rows(HID2, p[pre + "_f3"], (1, 1, R, 1, R), X32, ACT_NONE, X32)   # x + ffn(...): residual in the epilogue
        if not alias:
            ops.cast_rows(X32, CAT[:, :256])

    @torch.no_grad()
    def forward(self, data):
        for key in self.required_data_keys:
            assert key in data, f"Missing key {key} in data"
        kp0, kp1 = data["keypoints0"], data["keypoints1"]
        if not kp0.is_cuda:
            raise GimHipError("gim_amd LightGlue needs device (cuda/HIP) tensors: there is no CPU fallback")
        dev = kp0.device
        dt_want = {"bf16": GIM_BF16, "fp16": GIM_F16, "fp32": GIM_F32}[self.precision]
        if self._packed is None or self._packed[4] != dev or self._packed[3] != dt_want:
            self._prepack(dev)
        layers, head, wr, dt, _ = self._packed
        tdt = torch_dtype(dt)
        B, M, _ = kp0.shape
        N = kp1.shape[1]
        size0 = (data["image_size0"] if "image_size0" in data else data["resize0"])[:, [1, 0]]
        size1 = (data["image_size1"] if "image_size1" in data else data["resize1"])[:, [1, 0]]
        d0, d1 = data["descriptors0"], data["descriptors1"]
        assert d0.shape[-1] == self.conf["input_dim"] and d1.shape[-1] == self.conf["input_dim"]
        n_layers = self.conf["n_layers"]
        if M == 0 or N == 0:
            raise GimHipError("LightGlue needs at least one keypoint per image")
        enc0 = ops.lg_posenc(kp0.float().contiguous(), size0.to(device=dev, dtype=torch.float32).contiguous(), wr)
        enc1 = ops.lg_posenc(kp1.float().contiguous(), size1.to(device=dev, dtype=torch.float32).contiguous(), wr)
        enc = torch.cat([enc0, enc1], 0)
        R0, R1 = B * M, B * N
        R = R0 + R1
        alias = dt == GIM_F32
        CAT = torch.empty(R, 512, dtype=tdt, device=dev)
        X32 = CAT[:, :256] if alias else torch.empty(R, 256, dtype=torch.float32, device=dev)
        X32[:R0].copy_(d0.reshape(R0, 256))
        X32[R0:].copy_(d1.reshape(R1, 256))
        if not alias:
            ops.cast_rows(X32, CAT[:, :256])
        QKV = torch.empty(R, 768, dtype=tdt, device=dev)
        CTX = torch.empty(R, 256, dtype=tdt, device=dev)
        HID = torch.empty(R, 512, dtype=torch.float32, device=dev)
        HID2 = torch.empty(R, 512, dtype=tdt, device=dev)
        Mp, Np = (M + 63) // 64 * 64, (N + 63) // 64 * 64
        VT0 = torch.empty(B, 256, Mp, dtype=tdt, device=dev)
        VT1 = torch.empty(B, 256, Np, dtype=tdt, device=dev)
        x_t = CAT[:, :256]
        for p in layers:
            # ---- SelfBlock (lightglue.py:142-156) ----
            ops.linear(x_t, p["s_qkv"], QKV)
            ops.lg_rotary(QKV, enc, 512)
            ops.lg_transpose(QKV[:R0, 512:], VT0, B, M, Mp, 256)
            ops.lg_transpose(QKV[R0:, 512:], VT1, B, N, Np, 256)
            ops.sdpa(QKV[:R0, :256], QKV[:R0, 256:512], VT0, CTX[:R0], B, 4, M, M, Mp)
            ops.sdpa(QKV[R0:, :256], QKV[R0:, 256:512], VT1, CTX[R0:], B, 4, N, N, Np)
            ops.linear(CTX, p["s_out"], CAT[:, 256:])
            self._ffn(p, "s", CAT, X32, HID, HID2, alias)
            # ---- CrossBlock (lightglue.py:183-211): q and k share to_qk; both directions ----
            ops.linear(x_t, p["c_qkv"], QKV[:, :512])
            ops.lg_transpose(QKV[:R0, 256:512], VT0, B, M, Mp, 256)
            ops.lg_transpose(QKV[R0:, 256:512], VT1, B, N, Np, 256)
            ops.sdpa(QKV[:R0, :256], QKV[R0:, :256], VT1, CTX[:R0], B, 4, M, N, Np)
            ops.sdpa(QKV[R0:, :256], QKV[:R0, :256], VT0, CTX[R0:], B, 4, N, M, Mp)
            ops.linear(CTX, p["c_out"], CAT[:, 256:])
            self._ffn(p, "c", CAT, X32, HID, HID2, alias)
        # ---- MatchAssignment + filter_matches (lightglue.py:248-300) ----
        desc = X32.contiguous() if alias else X32
        MD = torch.empty(R, 256, dtype=torch.float32, device=dev)
        ops.linear(x_t, head["final"], MD)
        r = ops.lg_assign(desc[:R0].view(B, M, 256), desc[R0:].view(B, N, 256), MD[:R0].view(B, M, 256),
                          MD[R0:].view(B, N, 256), head["mw"], head["mb"], float(self.conf["filter_threshold"]))
        counts = r.count.tolist()  # the one read-back: sizes of the per-pair match lists (torch.where, lightglue.py:500)
        total = sum(counts)
        kscale = data.get("_adapter")  # set by gim_lightglue_inference: (scale0, scale1) -> fused caller-side adapter
        if kscale is not None:
            packed = ops.lg_emit_matches(r, total, kp0.float().contiguous(), kp1.float().contiguous(), kscale[0], kscale[1])
        else:
            packed = ops.lg_emit_matches(r, total)
        offs = [0]
        for cnt in counts:
            offs.append(offs[-1] + cnt)
        pred = {
            "matches0": r.matches0, "matches1": r.matches1,
            "matching_scores0": r.mscores0, "matching_scores1": r.mscores1,
            "ref_descriptors0": desc[:R0].view(B, 1, M, 256), "ref_descriptors1": desc[R0:].view(B, 1, N, 256),
            "log_assignment": LazyLogAssignment(r),
            "stop": n_layers,
            "matches": [packed[0][offs[b]:offs[b + 1]] for b in range(B)],
            "scores": [packed[1][offs[b]:offs[b + 1]] for b in range(B)],
            "prune0": torch.full((B, M), float(n_layers), device=dev), "prune1": torch.full((B, N), float(n_layers), device=dev),
        }
        if kscale is not None:
            pred["_packed"] = packed
        return pred
