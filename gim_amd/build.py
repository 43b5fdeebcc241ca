"""Builds libgimhip.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch involved.

    python -m gim_amd.build           # incremental
    python -m gim_amd.build --force
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgimhip.so")
SOURCES = ["runtime.hip", "conv_igemm.hip", "elementwise.hip", "linear_attention.hip", "coarse_match.hip",
           "fine_match.hip", "fine_fused.hip", "token_mlp.hip", "bneck_fused.hip", "bneck_tail.hip", "stem7x7.hip", "emit.hip", "superpoint.hip", "lightglue.hip", "lg_assign.hip", "dkm.hip", "gp_solve.hip", "sample.hip"]
# the gim_loftr path exists in two 16-bit operand flavours (bf16 / IEEE fp16, csrc/gim_common.h): these files are compiled a
# second time with -DGIM_HALF_KIND=1 into *_f16.o, whose entry points carry the suffix `_f16`
F16_SOURCES = ["conv_igemm.hip", "elementwise.hip", "linear_attention.hip", "coarse_match.hip", "fine_match.hip", "fine_fused.hip",
               "token_mlp.hip", "bneck_fused.hip", "bneck_tail.hip", "stem7x7.hip",
               "dkm.hip", "lightglue.hip", "superpoint.hip"]   # round 5: the dense matchers / gim_lightglue in the fp16 flavour too
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"] + os.environ.get("GIM_HIPCC_EXTRA", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "gim_hip.h"))
    return hdrs


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    # objects built with other flags (GIM_HIPCC_EXTRA: development builds) are stale
    stamp = os.path.join(objdir, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(FLAGS):
        force = True
    newest_hdr = max(os.path.getmtime(h) for h in _deps())
    jobs = []
    objs = []
    for src, extra, suffix in [(s_, [], ".o") for s_ in SOURCES] + [(s_, ["-DGIM_HALF_KIND=1"], "_f16.o") for s_ in F16_SOURCES]:
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.replace(".hip", suffix))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), newest_hdr):
            jobs.append([hipcc, *FLAGS, *extra, "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=min(int(os.environ.get("GIM_BUILD_JOBS", "6")), max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    with open(stamp, "w") as f:
        f.write(" ".join(FLAGS))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
