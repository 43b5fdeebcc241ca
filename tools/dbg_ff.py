import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
import torch
import test_gpu_fine_fused as T
from tools import synth_loftr as S
model, sd = S.synthetic_model("bf16"); model = model.cuda()
case = T._case(64, 1)
e_f, k_f, a0, a1 = T._run(model, case, True)
e_u, k_u, u0, u1 = T._run(model, case, False)
t0, t1, fm = T._oracle(sd, case)
for nm, a, u, t in (("fine0", a0, u0, t0), ("fine1", a1, u1, t1)):
    sc = t.abs().max()
    print(nm, "fused-unfused mean", ((a-u).abs().mean()/sc).item(), "fused-oracle", ((a-t).abs().mean()/sc).item(), "unfused-oracle", ((u-t).abs().mean()/sc).item())
    err = (a-t).abs()/sc
    print("  per head (16ch) mean err:", [round(x,4) for x in err.view(-1,25,8,16).mean((0,1,3)).tolist()])
    print("  per token mean err:", [round(x,4) for x in err.mean((0,2)).tolist()])
    print("  per match%4 mean err:", [round(err[i::4].mean().item(),4) for i in range(4)])
    print("  per 32ch group:", [round(x,4) for x in err.view(-1,25,4,32).mean((0,1,3)).tolist()])
