"""Static instruction mix of a kernel, basic block by basic block (no GPU: hipcc -S for gfx950 and a small parser).
For every block: loop depth (from the compiler's own loop comments), instruction counts by class and the issue cycles they stand for on
one SIMD -- MFMA by opcode (passes x 4), transcendentals 16, other VALU 4 (wave64 on a 16-lane SIMD), LDS / vector memory as issue slots.
The ratio MFMA cycles : VALU cycles : LDS operations of an inner loop is the static ceiling of the MFMA pipe's occupancy in that loop when
one wave owns the SIMD; with two waves the VALU work of one can hide under the MFMAs of the other.

    python tools/isa_mix.py gim_amd/csrc/token_mlp.hip token_mlp_kernel [--f16] [--min 24] [-D NAME=VALUE ...]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

# v_mfma_* passes (4 cycles each) on gfx950, by shape suffix; anything unknown counts 8 passes
MFMA_PASSES = {"32x32x16": 8, "16x16x32": 4, "32x32x8": 16, "16x16x16": 8, "32x32x2_f32": 16, "16x16x4_f32": 8, "32x32x1": 16,
               "16x16x4_f64": 8, "4x4x4": 2, "32x32x64": 16, "16x16x128": 8}
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        for k, p in MFMA_PASSES.items():
            if k in op:
                return "mfma", 4 * p
        return "mfma", 32
    if op.startswith("v_accvgpr"):
        return "valu", 4
    if op.startswith(TRANS):
        return "trans", 16
    if op.startswith("v_"):
        return "valu", 4
    if op.startswith("ds_"):
        return "lds", 0
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return ("scratch" if op.startswith("scratch_") else "vmem"), 0
    if op.startswith("s_waitcnt"):
        return "wait", 0
    if op.startswith("s_barrier"):
        return "barrier", 0
    if op.startswith("s_"):
        return "salu", 0
    return "other", 0


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("source")
    ap.add_argument("kernel", help="substring of the (mangled or demangled) kernel name")
    ap.add_argument("--f16", action="store_true", help="-DGIM_HALF_KIND=1 (the fp16 objects of the library)")
    ap.add_argument("--min", type=int, default=24, help="smallest block (instructions) to print")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-comment", "--cuda-device-only", "-S", os.path.join(ROOT, a.source), "-o", asm]
        cmd += ["-DGIM_HALF_KIND=1"] if a.f16 else []
        cmd += [f"-D{d}" for d in a.D]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        lines = open(asm).read().splitlines()
    # function bodies: "<mangled>:" ... "s_endpgm"
    names = [(i, m.group(1)) for i, ln in enumerate(lines) for m in [re.match(r"^(_Z\w+):", ln)] if m]
    filt = subprocess.run(["c++filt"] + [n for _, n in names], capture_output=True, text=True).stdout.splitlines()
    picks = [(i, n, d) for (i, n), d in zip(names, filt) if a.kernel in n or a.kernel in d]
    if not picks:
        sys.exit(f"no kernel matching {a.kernel!r}; have: " + ", ".join(sorted({d.split('(')[0][-60:] for d in filt}))[:1500])
    for start, mangled, dem in picks:
        print(f"== {dem.replace('(anonymous namespace)::', '')[:150]}")
        blocks, cur, depth = collections.OrderedDict(), "entry", 0
        blocks[cur] = [0, collections.Counter(), collections.Counter()]
        for ln in lines[start + 1:]:
            s = ln.strip()
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                cur = m.group(1)
                d = re.search(r"Depth[= ](\d+)", ln)
                depth = int(d.group(1)) if d else 0
                blocks[cur] = [depth, collections.Counter(), collections.Counter()]
                continue
            if s.startswith(";") and "in Loop" in s or (s.startswith(";") and "Loop Header" in s):
                d = re.search(r"Depth[= ](\d+)", s)
                if d:
                    blocks[cur][0] = max(blocks[cur][0], int(d.group(1)))
                continue
            if not s or s.startswith((";", ".", "//")):
                continue
            op = s.split()[0]
            cls, cyc = classify(op)
            blocks[cur][1][cls] += 1
            blocks[cur][2][cls] += cyc
            if op == "s_endpgm":
                break
        tot_n, tot_c = collections.Counter(), collections.Counter()
        print(f"{'block':12s} depth  {'mfma':>5s} {'(cyc)':>7s} {'valu':>5s} {'trans':>5s} {'(cyc)':>7s} {'lds':>4s} {'vmem':>4s} {'scr':>3s} {'wait':>4s} {'bar':>3s} {'salu':>5s}   MFMA : VALU cycles")
        for b, (depth, n, c) in blocks.items():
            tot_n.update(n)
            tot_c.update(c)
            size = sum(n.values())
            if size < a.min:
                continue
            vc = c["valu"] + c["trans"]
            ratio = f"{c['mfma'] / vc:5.2f}" if vc else "  inf"
            print(f"{b:12s} {depth:5d}  {n['mfma']:5d} {c['mfma']:7d} {n['valu']:5d} {n['trans']:5d} {vc:7d} {n['lds']:4d} {n['vmem']:4d} {n['scratch']:3d} {n['wait']:4d} "
                  f"{n['barrier']:3d} {n['salu']:5d}   {ratio}")
        vc = tot_c["valu"] + tot_c["trans"]
        print(f"{'static total':12s}        {tot_n['mfma']:5d} {tot_c['mfma']:7d} {tot_n['valu']:5d} {tot_n['trans']:5d} {vc:7d} {tot_n['lds']:4d} {tot_n['vmem']:4d} {tot_n['scratch']:3d} "
              f"{tot_n['wait']:4d} {tot_n['barrier']:3d} {tot_n['salu']:5d}")
        print()


if __name__ == "__main__":
    main()
