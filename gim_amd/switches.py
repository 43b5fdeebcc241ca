"""Development switches of the engine, in ONE place.

Every fused kernel of the gim_loftr / gim_dkm / gim_roma paths keeps the launch sequence it replaced as a cross-check (the tests compare
the two) and every launch-order experiment keeps its A/B.  Until round 4 each of them was an environment variable of its own (46 `GIM_*`
names); since round 5 they are plain attributes of the module objects, set

  * by the caller:  `config['fine_fused'] = False` (gim_loftr config dict), `model.bneck_tail = False` + `model._invalidate()`, ...
  * or, for A/B runs of unmodified callers, through the single environment variable
        GIM_FLAGS="fine_fused=0,tf_chains=4"        (separators: , ; +)
    read once at import.

Environment variables the package reads -- all of them, the C library reads none (round 6: the last two getenv() calls left csrc/):
GIM_PRECISION (default precision mode, gim_amd/precision.py), GIM_FLAGS (this file), GIM_LIB (an alternative libgimhip.so,
gim_amd/_lib.py), GIM_POSE_BACKEND (host RANSAC backend, gim_amd/pose.py) and the build's GIM_HIPCC_EXTRA / GIM_BUILD_JOBS
(gim_amd/build.py).

Diagnostics (round 6): a GIM_FLAGS key nobody asked for by the end of the process (a misspelt `fine_fuse=0` silently measured the
default twice) and any of the per-switch GIM_* variables of rounds 1-4 still set in the environment (GIM_GRAPH=0, GIM_FINE_FUSED=0, ...:
ignored since round 5) draw ONE warning each."""
import atexit
import os
import warnings


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
_ASKED = set()
_KNOWN_ENV = {"GIM_PRECISION", "GIM_FLAGS", "GIM_LIB", "GIM_POSE_BACKEND", "GIM_HIPCC_EXTRA", "GIM_BUILD_JOBS", "GIM_WEIGHTS_DIR", "GIM_ZEB_DIR",
              "GIM_SKIP_SLOW_TESTS", "GIM_CPU_THREADS", "GIM_AB_PRECISION"}


def legacy_env(environ=None):
    """GIM_* variables set in the environment that nothing reads any more (the per-switch variables of rounds 1-4); GIM_BENCH_* belong to bench.py"""
    environ = os.environ if environ is None else environ
    return sorted(k for k in environ if k.startswith("GIM_") and k not in _KNOWN_ENV and not k.startswith("GIM_BENCH_"))


def unused_flags():
    """GIM_FLAGS keys no flag() / tri_flag() call has asked for so far"""
    return sorted(k for k in FLAGS if k not in _ASKED)


_old = legacy_env()
if _old:
    warnings.warn(f"gim_amd: {', '.join(_old)} set in the environment but no longer read (since round 5 the switches are module attributes / "
                  f"config keys or GIM_FLAGS=\"name=value,...\", gim_amd/switches.py)", stacklevel=2)


@atexit.register
def _report_unused():
    left = unused_flags()
    if left:
        warnings.warn(f"gim_amd: GIM_FLAGS key(s) {left} were never consumed by this process -- misspelt, or a switch of a module that was not built")


def flag(name, default, config=None):
    """value of switch `name`: config[name] if the caller set it, else GIM_FLAGS, else `default` (whose type the result takes)"""
    _ASKED.add(name.lower())
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


def tri_flag(name, config=None):
    """three-state switch: None when unset (the module decides), else the boolean rule of flag() -- any spelling flag() accepts"""
    v = flag(name, "", config)
    if v is None or (isinstance(v, str) and v.strip() == ""):
        return None
    if isinstance(v, bool):
        return v
    t = str(v).strip().lower()
    if t in ("1", "true", "yes", "on"):
        return True
    if t in ("0", "false", "no", "off"):
        return False
    raise ValueError(f"switch {name}={v!r}: expected one of 1/0, true/false, yes/no, on/off (or unset)")
