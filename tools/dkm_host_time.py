import sys, time, torch
sys.path.insert(0, '/root/repo')
from gim_amd.dkm import DKMv3
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = DKMv3(None, 672, 896, upsample_preds=True, precision='bf16').eval()
im0 = torch.rand(1,3,480,640).to(dev); im1 = torch.rand(1,3,480,640).to(dev)
for _ in range(2): m.match(im0, im1)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); m.match(im0, im1); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host issue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
