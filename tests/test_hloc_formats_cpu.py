"""CPU: gim_amd/hloc_formats.py against the vectors oracle/make_golden_hloc.py recorded by executing the reference's own
`hloc/match_dense.py` helpers -- ids, matches0 and the voted keypoints are exact."""
import os
from collections import Counter

import numpy as np

import make_golden_hloc as G
from gim_amd import hloc_formats as H


class _FakeH5(dict):
    """the slice of h5py's group protocol the writers use"""

    def create_group(self, name):
        self[name] = _FakeH5()
        return self[name]

    def create_dataset(self, name, data):
        self[name] = np.asarray(data)


def test_aggregation_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "hloc_formats.npz"))
    pairs, data = G.scenario()
    imgs = {n: H.ImageKeypoints() for n in "ABC"}
    left = Counter(n for p in pairs for n in p)
    final = {}
    for n0, n1 in pairs:
        kp0, kp1, sc = data[(n0, n1)]
        tag = n0 + n1
        assert np.array_equal(kp0, g[f"{tag}_kp0"]) and np.array_equal(sc, g[f"{tag}_scores"])     # same scenario
        ids0 = imgs[n0].add(kp0, sc, 2, 8)
        ids1 = imgs[n1].add(kp1, sc, 2, 8)
        assert np.array_equal(ids0, g[f"{tag}_ids0"]) and np.array_equal(ids1, g[f"{tag}_ids1"])
        m0, s0 = H.matches0_from_ids(ids0, ids1, sc)
        assert m0.dtype == np.int32 and s0.dtype == np.float16
        assert np.array_equal(m0, g[f"{tag}_matches0"]) and np.array_equal(s0, g[f"{tag}_scores0"])
        for name in (n0, n1):
            left[name] -= 1
            if left[name] == 0:
                final[name] = imgs[name].finalize()
                assert final[name][0].dtype == np.float32
                assert np.array_equal(final[name][0], g[f"{name}_keypoints"])
                assert np.array_equal(final[name][1], g[f"{name}_kp_score"])
    for n0, n1 in pairs:                                     # assign_matches: nearest final keypoint within max_error
        kp0, kp1, sc = data[(n0, n1)]
        ids0 = H.nearest_ids(kp0, final[n0][0], 2)
        ids1 = H.nearest_ids(kp1, final[n1][0], 2)
        assert np.array_equal(ids0, g[f"{n0}{n1}_nn_ids0"])
        m0, s0 = H.matches0_from_ids(ids0, ids1, sc)
        assert np.array_equal(m0, g[f"{n0}{n1}_nn_matches0"]) and np.array_equal(s0, g[f"{n0}{n1}_nn_scores0"])
    pts = data[pairs[0]][0][:50]
    for cell, key in ((8, "cpts8"), (2, "cpts2"), (0.0, "cpts0")):
        assert np.array_equal(np.asarray(H.quantize(pts, cell), dtype=np.float64), g[key])


def test_edge_cases_and_writers():
    m0, s0 = H.matches0_from_ids(np.array([-1, -1]), np.array([0, 1]), np.array([0.5, 0.7], dtype=np.float32))
    assert m0.shape == (0,) and m0.dtype == np.int32 and s0.dtype == np.float16
    assert H.nearest_ids(np.zeros((3, 2)), np.zeros((0, 2)), 2).shape == (0,)
    # n-to-1 on both sides: only pairs that are the best of their keypoint on BOTH sides survive
    m0, s0 = H.matches0_from_ids(np.array([0, 0, 1, 2]), np.array([5, 6, 6, 7]), np.array([0.9, 0.2, 0.8, 0.4], dtype=np.float32))
    assert m0.tolist() == [5, 6, 7] and np.allclose(s0.astype(np.float32), [0.9, 0.8, 0.4], atol=1e-3)
    top = H.ImageKeypoints()
    top.add(np.array([[10.2, 10.1], [40.0, 40.0], [10.4, 10.3]], dtype=np.float32), np.array([0.3, 0.9, 0.4], dtype=np.float32))
    kps, score = top.finalize(max_kps=1)
    assert kps.shape == (1, 2) and np.isclose(score[0], 0.9)
    fd = _FakeH5()
    assert H.pair_key("db/a.jpg", "q/b.jpg") == "db-a.jpg/q-b.jpg"
    grp = H.write_dense_pair(fd, "db/a.jpg", "q/b.jpg", np.zeros((4, 2)), np.ones((4, 2)), np.ones(4))
    H.write_matches0(grp, [1, -1, 0], [0.5, 0.0, 0.25])
    H.write_matches0(grp, [2, -1, 0], [0.5, 0.0, 0.25])                      # overwrite like assign_matches does
    assert grp["keypoints0"].dtype == np.float32 and grp["matches0"].dtype == np.int32 and grp["matches0"][0] == 2
    assert grp["matching_scores0"].dtype == np.float16
    sp = H.write_sparse_matches(fd, "a", "b", np.array([3, -1], dtype=np.int64), np.array([0.7, 0.0]))
    assert sp["matches0"].dtype == np.int16 and sp["matching_scores0"].dtype == np.float16
    kg = H.write_keypoints(fd, "a", kps, score)
    assert kg["keypoints"].dtype == np.float32


def test_writers_through_a_real_h5py_file(tmp_path):
    """the wire formats written through h5py itself (skipped where h5py is not installed): group layout,
    dtypes and the replace-on-rewrite behaviour of match_dense.py:248-257,353-356,375-381 / match_features.py:150-160"""
    import pytest
    h5py = pytest.importorskip("h5py")
    rng = np.random.default_rng(3)
    kp0, kp1, sc = rng.random((50, 2)) * 640, rng.random((50, 2)) * 480, rng.random(50)
    path = str(tmp_path / "matches.h5")
    with h5py.File(path, "a", libver="latest") as fd:
        grp = H.write_dense_pair(fd, "db/a.jpg", "q/b.jpg", kp0, kp1, sc)
        H.write_matches0(grp, np.arange(50) % 7 - 1, sc)
        H.write_dense_pair(fd, "db/a.jpg", "q/b.jpg", kp0[:10], kp1[:10], sc[:10])           # same pair again: replaced, not duplicated
        H.write_sparse_matches(fd, "db/a.jpg", "q/c.jpg", np.arange(2048) % 300 - 1, rng.random(2048))
        H.write_keypoints(fd, "db/a.jpg", kp0, sc)
    with h5py.File(path, "r") as fd:
        g = fd[H.pair_key("db/a.jpg", "q/b.jpg")]
        assert g["keypoints0"].shape == (10, 2) and g["keypoints0"].dtype == np.float32 and g["scores"].dtype == np.float32
        assert "matches0" not in g                                                              # the rewrite started a fresh group
        s = fd[H.pair_key("db/a.jpg", "q/c.jpg")]
        assert s["matches0"].dtype == np.int16 and s["matching_scores0"].dtype == np.float16 and s["matches0"].shape == (2048,)
        k = fd["db/a.jpg"]
        assert k["keypoints"].dtype == np.float32 and np.allclose(k["keypoints"][()], kp0.astype(np.float32))
