"""Stage-by-stage check of fine_fused.hip built with -DFF_DEBUG_STAGES (GIM_HIPCC_EXTRA): dumps of the LDS tiles after
each step of each encoder call against a plain torch fp32 restatement of transformer.py:35-58 on the same inputs."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch
import torch.nn.functional as F
import test_gpu_fine_fused as T
import loftr_oracle as O
from tools import synth_loftr as S

model, sd = S.synthetic_model("bf16"); model = model.cuda()
case = T._case(64, 1)
f0, f1, b, i, j, mk1c = case
n0, n1 = f0.float().permute(0, 3, 1, 2).contiguous(), f1.float().permute(0, 3, 1, 2).contiguous()
w0, w1 = O.fine_preprocess(n0, n1, b, i, j, (12, 16), (48, 64), 5)   # [M,25,128] fp32 (bf16-valued)
bf = lambda t: t.to(torch.bfloat16).float()

def layer_stages(p, x, src):
    W = lambda k: bf(sd[f"{p}.{k}.weight"])
    st = {}
    q = F.elu(x @ W("q_proj").T) + 1; k = F.elu(src @ W("k_proj").T) + 1; v = src @ W("v_proj").T
    st[1], st[2], st[3] = bf(k), bf(v), bf(q)
    M = x.shape[0]
    Q, K, V = st[3].view(M, 25, 8, 16), st[1].view(M, 25, 8, 16), st[2].view(M, 25, 8, 16)   # already elu+1 / bf16
    KV = torch.einsum("nshd,nshv->nhdv", K, V)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + 1e-6)
    msg = (torch.einsum("nlhd,nhdv->nlhv", Q, KV) * Z[..., None]).reshape(M, 25, 128)
    st[4] = bf(msg)
    m = st[4] @ W("merge").T
    m = F.layer_norm(m, (128,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    st[5] = bf(m)
    h = F.relu(torch.cat([bf(x), st[5]], 2) @ W("mlp.0").T)
    st[6], st[7] = bf(h[..., :128]), bf(h[..., 128:])
    o = bf(h) @ W("mlp.2").T
    o = F.layer_norm(o, (128,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    xn = x + o
    st[8] = bf(xn)
    return st, xn

calls = []
x0, x1 = w0, w1
s, x0 = layer_stages("loftr_fine.layers.0", x0, x0); calls.append(s)
s, x1 = layer_stages("loftr_fine.layers.0", x1, x1); calls.append(s)
s, x0 = layer_stages("loftr_fine.layers.1", x0, bf(x1)); calls.append(s)
s, x1 = layer_stages("loftr_fine.layers.1", x1, bf(x0)); calls.append(s)
names = {1: "K", 2: "V", 3: "Q", 4: "msg", 5: "LN1", 6: "hid_lo", 7: "hid_hi", 8: "x_out"}
for c in range(4):
    for sid in range(1, 9):
        os.environ["GIM_FF_STAGE"] = str(c * 10 + sid)
        _, _, d0, _ = T._run(model, case, True)
        ref = calls[c][sid]
        err = (d0 - ref).abs()
        print(f"call {c} stage {sid} {names[sid]:7s} mean {err.mean().item() / ref.abs().max().item():.5f} max {err.max().item() / ref.abs().max().item():.4f}  (ref scale {ref.abs().max().item():.2f})", flush=True)
os.environ["GIM_FF_STAGE"] = "4"
_, _, d0, _ = T._run(model, case, True)
ref = calls[0][4]
err = (d0 - ref).abs() / ref.abs().max()
print("per match:", [round(x, 3) for x in err.mean((1, 2)).tolist()])
print("per head:", [round(x, 4) for x in err.view(-1, 25, 8, 16).mean((0, 1, 3)).tolist()])
print("per token:", [round(x, 4) for x in err.mean((0, 2)).tolist()])
bad = err.mean((1,2)).argmax().item()
print("worst match", bad, "per head", [round(x, 3) for x in err[bad].view(25, 8, 16).mean((0, 2)).tolist()], "i,j", case[2][bad].item(), case[3][bad].item())
print("ratio kernel/ref worst match head0 tok0..4:", (d0[bad, :5, :4] / ref[bad, :5, :4]).tolist())
