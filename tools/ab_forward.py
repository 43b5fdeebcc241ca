"""A/B timing of the benchmarked gim_loftr step with two builds of libgimhip.so in ONE gpurun call (box clocks drift between
calls): python tools/ab_forward.py [reps]   ->  alternates `GIM_LIB=a` / `GIM_LIB=b` sub-processes, prints ms/step of each.
The alternative library is gim_amd/lib/alt/libgimhip.so (build the other revision there by hand)."""
import os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    import torch
    sys.path.insert(0, root)
    from tools import synth_loftr as S
    m, _ = S.synthetic_model(os.environ.get("GIM_AB_PRECISION", "fp16"))
    m = m.cuda()
    c0, c1 = S.textured_pairs(int(os.environ.get("GIM_AB_BATCH", "8")), 480, 640, seed=1234, frac=0.45)
    c0, c1 = c0.cuda(), c1.cuda()
    def step():
        d = {"image0": c0[:, :1], "image1": c1[:, :1], "color0": c0, "color1": c1}
        m(d)
        return d
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            d = step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    print(f"{best:.3f}")
    sys.exit(0)
# usage: ab_forward.py [reps] [variant ...]; a variant is `lib=alt` (gim_amd/lib/alt/libgimhip.so), `lib=new`, or ENV=value[,ENV=value]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
variants = sys.argv[2:] or ["lib=alt", "lib=new"]
lib, alt, keep = (os.path.join(root, "gim_amd", "lib", x) for x in ("libgimhip.so", "alt/libgimhip.so", "libgimhip.keep"))
subprocess.run(["cp", lib, keep], check=True)
res = {v: [] for v in variants}
try:
    for _ in range(reps):
        for v in variants:
            env = dict(os.environ)
            src = keep
            for kv in v.split(","):
                k, _, val = kv.partition("=")
                if k == "lib":
                    src = alt if val == "alt" else keep
                else:
                    env[k] = val
            subprocess.run(["cp", src, lib], check=True)
            out = subprocess.run([sys.executable, __file__, "--worker"], capture_output=True, text=True, env=env)
            res[v].append(out.stdout.strip().splitlines()[-1] if out.returncode == 0 else "ERR " + out.stderr[-300:])
finally:
    os.replace(keep, lib)
print(res)
