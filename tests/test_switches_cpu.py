"""gim_amd/switches.py: the one place development switches come from (config key > GIM_FLAGS > default), and the rule that the package reads
no other per-feature environment variable (46 GIM_* names -> one hook)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flag_resolution_order_and_types(monkeypatch):
    from gim_amd import switches
    monkeypatch.setattr(switches, "FLAGS", switches._parse("fine_fused=0, tf_chains=4;graph_cache=2+stem_split=off,force_big_tile"))
    assert switches.flag("fine_fused", True) is False and switches.flag("stem_split", True) is False
    assert switches.flag("tf_chains", 2) == 4 and switches.flag("graph_cache", 4) == 2
    assert switches.flag("force_big_tile", False) is True          # a bare name means "on"
    assert switches.flag("bneck_tail", True) is True                # absent: the default
    assert switches.flag("tf_chains", 2, {"tf_chains": 1}) == 1     # the caller's config wins over GIM_FLAGS
    assert switches.flag("fine_fused", True, {"fine_fused": None}) is False   # None = not set by the caller
    assert switches.flag("coarse_sim", "") == ""


def test_split_stem_default_follows_the_precision_mode():
    """`stem_split` = 'auto' (default): split-operand first convolution in the fp16 mode only; True / False (config, attribute or GIM_FLAGS
    text) force it; never in the fp32 mode."""
    from types import SimpleNamespace as NS
    from gim_amd.loftr.loftr import LoFTR
    sp = lambda prec, s: LoFTR._split(NS(precision=prec, stem_split=s))   # noqa: E731
    assert sp("fp16", "auto") is True and sp("bf16", "auto") is False and sp("fp32", "auto") is False
    assert sp("bf16", True) is True and sp("bf16", "1") is True and sp("fp16", False) is False and sp("fp16", "off") is False
    assert sp("fp32", True) is False


def test_package_reads_only_the_documented_environment_variables():
    allowed = {"GIM_PRECISION", "GIM_FLAGS", "GIM_LIB", "GIM_POSE_BACKEND", "GIM_HIPCC_EXTRA", "GIM_BUILD_JOBS", "GIMRECONSTRUCTION", "HIPCC"}
    seen = set()
    for dp, _, fs in os.walk(os.path.join(ROOT, "gim_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                seen |= set(re.findall(r"environ(?:\.get)?\(\s*[\"']([A-Z_0-9]+)[\"']", src)) | set(re.findall(r"environ\[\s*[\"']([A-Z_0-9]+)[\"']", src))
            if f.endswith((".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                seen |= set(re.findall(r"getenv\(\s*\"([A-Z_0-9]+)\"", src))
    extra = seen - allowed
    assert not extra, f"undocumented environment switches: {sorted(extra)}"


def test_the_library_reads_no_environment_variable():
    """round 6: csrc/ has no getenv() left (the pre-candidate capacity is gim_coarse_args.precand_per_row, the fine kernel's debug stage a -D build)"""
    for f in os.listdir(os.path.join(ROOT, "gim_amd", "csrc")):
        assert "getenv(" not in open(os.path.join(ROOT, "gim_amd", "csrc", f)).read(), f


def test_unknown_flags_and_legacy_variables_are_reported(monkeypatch):
    from gim_amd import switches
    monkeypatch.setattr(switches, "FLAGS", switches._parse("fine_fuse=0,tf_chains=4"))
    monkeypatch.setattr(switches, "_ASKED", set())
    switches.flag("tf_chains", 2)
    assert switches.unused_flags() == ["fine_fuse"]
    assert switches.legacy_env({"GIM_GRAPH": "0", "GIM_FLAGS": "x", "GIM_BENCH_DEBUG": "1", "GIM_LA_KV2": "1", "PATH": ""}) == ["GIM_GRAPH", "GIM_LA_KV2"]
    assert switches.tri_flag("gp_exact") is None
    monkeypatch.setattr(switches, "FLAGS", switches._parse("gp_exact=true,refiner=off"))
    assert switches.tri_flag("gp_exact") is True and switches.tri_flag("refiner") is False
    monkeypatch.setattr(switches, "FLAGS", switches._parse("gp_exact=maybe"))
    import pytest
    with pytest.raises(ValueError):
        switches.tri_flag("gp_exact")
