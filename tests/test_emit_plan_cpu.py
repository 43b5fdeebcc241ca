"""The launch plan of the coarse transformer with projections emitted by the token tails (gim_amd/loftr/loftr.py::_emit_plan,
transformer.py:80-101): every call must find the q projection of its query rows and the k / v projections of its source rows computed
from the CURRENT value of those rows -- produced exactly once, by the initial GEMMs or by the tail that last updated the rows."""
import itertools

import pytest
import torch

import gim_amd.loftr.loftr as L


class _Proj:
    def __init__(self, tag):
        self.weight = torch.full((256, 256), float(tag))


class _Layer:
    def __init__(self, li):
        self.q_proj, self.k_proj, self.v_proj = _Proj(3 * li), _Proj(3 * li + 1), _Proj(3 * li + 2)


class _TF:
    def __init__(self, names):
        self.layer_names = list(names)
        self.layers = [_Layer(i) for i in range(len(names))]


SEQS = [["self", "cross"] * 4, ["cross", "self"] * 2, ["self", "self", "cross", "cross"], ["cross"], ["self"],
        ["cross", "cross", "cross"], ["self", "cross", "cross", "self", "self"]]


@pytest.mark.parametrize("emit_q", [True, False], ids=["q-emitted", "q-local"])
@pytest.mark.parametrize("names,same_len", list(itertools.product(SEQS, [True, False])),
                         ids=[f"{'-'.join(n[0] for n in s)}-{'eq' if e else 'ne'}" for s, e in itertools.product(SEQS, [True, False])])
def test_every_projection_is_fresh_and_produced_once(monkeypatch, names, same_len, emit_q):
    """emit_q = False (round 5, `q_local`): every call projects its own queries inside the token kernel -- the plan holds k / v blocks only"""
    packed = []
    monkeypatch.setattr(L, "pack_token_emit", lambda ws, dev, tdt: packed.append([int(w[0, 0]) for w in ws]) or len(packed) - 1)
    calls, per_call, initial = L.LoFTR._emit_plan(_TF(names), same_len, "cpu", torch.float16, emit_q=emit_q)
    version = {0: 0, 1: 0}
    have = {}          # (layer, block, side) -> version of the rows it was computed from
    produced = set()
    for li, blk, sides in initial:
        for sd in sides:
            assert (li, blk, sd) not in produced
            produced.add((li, blk, sd))
            have[(li, blk, sd)] = 0
    expect = []
    for li, kind in enumerate(names):
        if kind == "self":
            expect += [(li, (0, 1), (0, 1))] if same_len else [(li, (0,), (0,)), (li, (1,), (1,))]
        else:
            expect += [(li, (0,), (1,)), (li, (1,), (0,))]
    assert calls == expect
    for (li, xs, ss), em in zip(calls, per_call):
        for sd in xs:
            assert not emit_q or have.get((li, 0, sd)) == version[sd], ("stale / missing q", li, sd)
        for sd in ss:
            for blk in (1, 2):
                assert have.get((li, blk, sd)) == version[sd], ("stale / missing k, v", li, blk, sd)
        for sd in xs:
            version[sd] += 1
        if em is None:
            continue
        handle, blocks = em
        assert 0 < len(blocks) <= 6
        assert packed[handle] == [3 * l2 + blk for l2, blk, _ in blocks]      # the right weight matrix per block, in block order
        for l2, blk, sides in blocks:
            assert l2 >= li and set(sides) <= set(xs)
            for sd in sides:
                assert (l2, blk, sd) not in produced, ("projection computed twice", l2, blk, sd)
                produced.add((l2, blk, sd))
                have[(l2, blk, sd)] = version[sd]
    # nothing is computed that no call reads
    used = set()
    for li, xs, ss in calls:
        used |= ({(li, 0, sd) for sd in xs} if emit_q else set()) | {(li, b, sd) for sd in ss for b in (1, 2)}
    assert produced == used


@pytest.mark.parametrize("emit_q", [True, False], ids=["q-emitted", "q-local"])
@pytest.mark.parametrize("names,same_len", list(itertools.product(SEQS, [True, False])),
                         ids=[f"{'-'.join(n[0] for n in s)}-{'eq' if e else 'ne'}" for s, e in itertools.product(SEQS, [True, False])])
def test_fused_kv_pairs_cover_their_consumer_exactly(monkeypatch, names, same_len, emit_q):
    """_kv_consumers (round 5: token tails hand the k / v rows of the next attention over as partial KV states): a consuming call either gets
    its whole source from fused pairs -- every source side exactly once, from the tail that LAST updated that side -- or none of it; calls
    fed by the initial projections are never fused."""
    monkeypatch.setattr(L, "pack_token_emit", lambda ws, dev, tdt: 0)
    calls, per_call, initial = L.LoFTR._emit_plan(_TF(names), same_len, "cpu", torch.float16, emit_q=emit_q)
    cons = L.LoFTR._kv_consumers(calls, per_call, initial)
    fed_initial = {(li, sd) for li, blk, sides in initial if blk == 1 for sd in sides}
    covered = {}
    for (ci, bi), cj in cons.items():
        l2, blk, sides = per_call[ci][1][bi]
        assert blk == 1 and per_call[ci][1][bi + 1] == (l2, 2, sides) and cj > ci and calls[cj][0] == l2
        for sd in sides:
            assert sd in calls[ci][1] and sd in calls[cj][2]                 # produced by a call that updates the side, consumed as a source
            assert not any(sd in calls[c][1] for c in range(ci + 1, cj))      # ... and nobody updates the side in between
            assert (cj, sd) not in covered
            covered[(cj, sd)] = ci
    for cj, (li, xs, ss) in enumerate(calls):
        got = [sd for sd in ss if (cj, sd) in covered]
        assert got == [] or got == list(ss), ("a call's source is fused for some sides only", cj)
        if any((li, sd) in fed_initial for sd in ss):
            assert got == []
    # the benchmark's plan: everything but the first layer's calls is fused
    if names == ["self", "cross"] * 4 and not same_len:
        assert sorted({cj for cj in cons.values()}) == list(range(2, len(calls)))
