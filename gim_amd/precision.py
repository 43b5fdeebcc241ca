"""One place that resolves the `precision` of an engine (config value > GIM_PRECISION > the engine's default).

gim_loftr has three modes -- 'fp16' (its default), 'bf16', 'fp32'.  The secondary engines (SuperPoint, LightGlue, DKMv3, RoMa)
have the same three since round 5 (their kernels are compiled in the IEEE-fp16 flavour too: 11 instead of 8 significand bits per stored
activation); SuperPoint / LightGlue / DKMv3 default to 'bf16' -- ResNet-50's un-normalised residual streams carry no range guard there -- and a
process-wide GIM_PRECISION=fp16 set for gim_loftr still means 'bf16' to them: only an EXPLICIT precision='fp16' selects it.  gim_roma
defaults to 'fp16' since round 6 (`default="fp16"`): every 16-bit activation it stores sits behind a BatchNorm / LayerNorm (VGG19-BN, the
ViT's fp32 residual stream, the refiners' conv + BN blocks), its bf16 mode moves the warp by ~3 px at 560 x 560 (flipped anchor arg-max
decisions) where fp16 stays < 5e-6 of scale, and match() checks its outputs for non-finite values (gim_amd/roma/roma.py).  An unknown
value raises instead of silently selecting the fp32 path (five times slower)."""
import os

LOFTR_MODES = ("fp16", "bf16", "fp32")


def resolve(value, engine, default="bf16", env=True):
    """engine: 'loftr' (three modes) or anything else (two modes).  Returns the mode string."""
    explicit = value is not None
    p = value if explicit else (os.environ.get("GIM_PRECISION") if env else None)
    p = (p or default).lower()
    if engine == "loftr":
        if p not in LOFTR_MODES:
            raise ValueError(f"precision must be 'bf16', 'fp16' or 'fp32', got {p!r}")
        return p
    if p == "fp16" and not explicit and default != "fp16":
        return "bf16"      # GIM_PRECISION=fp16 set for gim_loftr: this engine's default 16-bit mode
    if p not in LOFTR_MODES:
        raise ValueError(f"{engine}: precision must be 'bf16', 'fp16' or 'fp32', got {p!r}")
    return p
