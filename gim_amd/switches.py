"""Development switches of the engine, in ONE place.

Every fused kernel of the gim_loftr / gim_dkm / gim_roma paths keeps the launch sequence it replaced as a cross-check (the tests compare
the two) and every launch-order experiment keeps its A/B.  Until round 4 each of them was an environment variable of its own (46 `GIM_*`
names); since round 5 they are plain attributes of the module objects, set

  * by the caller:  `config['fine_fused'] = False` (gim_loftr config dict), `model.bneck_tail = False` + `model._invalidate()`, ...
  * or, for A/B runs of unmodified callers, through the single environment variable
        GIM_FLAGS="fine_fused=0,tf_chains=4"        (separators: , ; +)
    read once at import.

Environment variables the package still reads: GIM_PRECISION (default precision mode, gim_amd/precision.py), GIM_FLAGS (this file),
GIM_LIB (an alternative libgimhip.so, gim_amd/_lib.py), GIM_POSE_BACKEND (host RANSAC backend, gim_amd/pose.py) and the build's
GIM_HIPCC_EXTRA / GIM_BUILD_JOBS (gim_amd/build.py)."""
import os


def _parse(text):
    out = {}
    for item in (text or "").replace(";", ",").replace("+", ",").split(","):
        item = item.strip()
        if not item:
            continue
        k, _, v = item.partition("=")
        out[k.strip().lower()] = v.strip() if _ else "1"
    return out


FLAGS = _parse(os.environ.get("GIM_FLAGS"))


def flag(name, default, config=None):
    """value of switch `name`: config[name] if the caller set it, else GIM_FLAGS, else `default` (whose type the result takes)"""
    v = None
    if config is not None:
        v = config.get(name)
    if v is None:
        v = FLAGS.get(name.lower())
    if v is None:
        return default
    if isinstance(default, bool):
        return (str(v).strip().lower() not in ("0", "false", "no", "off", "")) if not isinstance(v, bool) else v
    if isinstance(default, int):
        return int(v)
    return v
