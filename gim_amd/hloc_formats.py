"""hloc wire formats for the matchers' outputs, host side, numpy only.

What the reference's SfM / localisation glue stores and how it turns dense matches into keypoint-indexed matches:

  dense pair group      `keypoints0`, `keypoints1` fp32 [n,2], `scores` fp32 [n]                 hloc/match_dense.py:248-257
  aggregation           quantise keypoints to cells, vote for the sub-cell position, one keypoint per cell
                                                                                                hloc/match_dense.py:43-85, 298-390
  keypoint-indexed pair `matches0` int32 [K0] (-1 = unmatched), `matching_scores0` fp16 [K0]      hloc/match_dense.py:111-130
  re-assignment         nearest final keypoint within max_error                                  hloc/match_dense.py:58-63, 393-418
  sparse matcher output `matches0` int16, `matching_scores0` fp16                                hloc/match_features.py:150-160

h5py is not a dependency: the writers take any object with h5py's `create_group` / `create_dataset` / `in` / `del` protocol
(an `h5py.File` where it is installed).  Everything is deterministic and follows the reference's conventions for ties: ids are
handed out in order of first appearance, the winning sub-cell position of a cell is the first one that reached the top score
sum, score sums are accumulated in fp32 in arrival order.
"""
import numpy as np


def pair_key(name0, name1, separator="/"):
    """hloc/utils/parsers.py:51-52"""
    return separator.join((name0.replace("/", "-"), name1.replace("/", "-")))


def quantize(kpts, cell):
    """centre of the `cell`-sized patch a keypoint falls into (pixel-centre convention: +0.5 / -0.5), 2 decimals; cell <= 0 keeps
    the keypoints as they are (match_dense.py:43-46)"""
    kpts = np.asarray(kpts)
    if cell > 0.0:
        kpts = np.round(np.round((kpts + 0.5) / cell) * cell - 0.5, 2)
    return kpts


class ImageKeypoints:
    """Keypoints of one image while dense matches of several pairs are aggregated (`assign_keypoints(update=True)` +
    the finalisation of `aggregate_matches`)."""

    def __init__(self):
        self.cells = {}       # quantised cell position -> keypoint id (first appearance order)
        self.votes = []       # per keypoint id: {sub-cell position: fp32 score sum}, insertion ordered

    def __len__(self):
        return len(self.votes)

    def add(self, kpts, scores=None, max_error=2, cell_size=8):
        """ids [n] of the cells the keypoints fall into; unseen cells get new ids; every keypoint votes with its score for its
        position quantised to int(max_error)"""
        patch = max(cell_size if cell_size is not None else max_error, max_error)
        cell_pos = quantize(kpts, patch)
        vote_pos = quantize(kpts, int(max_error))
        ids = np.empty(len(cell_pos), dtype=np.int64)
        for n, (cp, vp) in enumerate(zip(map(tuple, cell_pos), map(tuple, vote_pos))):
            kid = self.cells.get(cp)
            if kid is None:
                kid = self.cells[cp] = len(self.votes)
                self.votes.append({})
            w = scores[n] if scores is not None else 1
            tally = self.votes[kid]
            tally[vp] = tally.get(vp, 0) + w
            ids[n] = kid
        return ids

    def finalize(self, max_kps=None):
        """(keypoints fp32 [K,2], score [K]): per cell the sub-cell position with the largest score sum (first one on ties);
        optionally only the max_kps best cells (ids of earlier `add` calls then refer to the un-truncated list:
        re-assign with `nearest_ids`, like the reference does)"""
        best = [max(t.items(), key=lambda kv: kv[1]) for t in self.votes]
        kps = np.array([b[0] for b in best], dtype=np.float32).reshape(-1, 2)
        score = np.array([b[1] for b in best], dtype=np.float64)
        if max_kps:
            top = np.argsort(score)[::-1][:min(max_kps, len(kps))]
            kps, score = kps[top], score[top]
        return kps, score


def nearest_ids(kpts, keypoints, max_error):
    """id of the nearest of `keypoints` within max_error px, else -1 (`assign_keypoints(update=False)`)"""
    from scipy.spatial import KDTree
    if len(keypoints) == 0:
        return np.array([], dtype=np.int64)
    dist, ids = KDTree(np.asarray(keypoints)).query(kpts)
    ids = np.asarray(ids).copy()
    ids[dist > max_error] = -1
    return ids


def _best_per_group(keys, scores):
    """index of the best-scoring element of every group of equal keys"""
    order = np.argsort(keys)
    bounds = np.flatnonzero(np.diff(keys[order])) + 1
    return [grp[np.argmax(scores[grp])] for grp in np.split(order, bounds) if len(grp)]


def matches0_from_ids(ids0, ids1, scores):
    """(matches0 int32 [max id0 + 1], matching_scores0 fp16): pairs with both ids valid, reduced to one-to-one by keeping a pair
    only if it is the best-scoring pair of its keypoint on BOTH sides (kpids_to_matches0 / get_unique_matches /
    matches_to_matches0, match_dense.py:88-130)"""
    ids0, ids1, scores = np.asarray(ids0), np.asarray(ids1), np.asarray(scores)
    valid = (ids0 != -1) & (ids1 != -1)
    a, b, s = ids0[valid], ids1[valid], scores[valid]
    if len(a) == 0:
        return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float16)
    keep = np.array(sorted(set(_best_per_group(a, s)) & set(_best_per_group(b, s))), dtype=np.int64)
    a, b, s = a[keep], b[keep], s[keep]
    if len(a) == 0:
        return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float16)
    matches0 = -np.ones(int(a.max()) + 1)
    scores0 = np.zeros(int(a.max()) + 1)
    matches0[a] = b
    scores0[a] = s
    return matches0.astype(np.int32), scores0.astype(np.float16)


def _replace(group, name, data):
    if name in group:
        del group[name]
    group.create_dataset(name, data=data)


def write_dense_pair(fd, name0, name1, kpts0, kpts1, scores):
    """match_dense.py:248-257: one group per pair with the raw dense matches (pixel coordinates of the original images)"""
    key = pair_key(name0, name1)
    if key in fd:
        del fd[key]
    grp = fd.create_group(key)
    grp.create_dataset("keypoints0", data=np.asarray(kpts0, dtype=np.float32))
    grp.create_dataset("keypoints1", data=np.asarray(kpts1, dtype=np.float32))
    grp.create_dataset("scores", data=np.asarray(scores, dtype=np.float32))
    return grp


def write_matches0(grp, matches0, scores0):
    """match_dense.py:353-356 / 414-418: keypoint-indexed matches of a dense pair group"""
    _replace(grp, "matches0", np.asarray(matches0, dtype=np.int32))
    _replace(grp, "matching_scores0", np.asarray(scores0, dtype=np.float16))


def write_sparse_matches(fd, name0, name1, matches0, matching_scores0=None):
    """match_features.py:150-160 (gim_lightglue through hloc): `matches0` int16, `matching_scores0` fp16"""
    key = pair_key(name0, name1)
    if key in fd:
        del fd[key]
    grp = fd.create_group(key)
    grp.create_dataset("matches0", data=np.asarray(matches0).astype(np.int16))
    if matching_scores0 is not None:
        grp.create_dataset("matching_scores0", data=np.asarray(matching_scores0).astype(np.float16))
    return grp


def write_keypoints(fd, name, keypoints, score):
    """match_dense.py:375-381: the aggregated keypoints of one image"""
    if name in fd:
        del fd[name]
    grp = fd.create_group(name)
    grp.create_dataset("keypoints", data=np.asarray(keypoints, dtype=np.float32))
    grp.create_dataset("score", data=np.asarray(score))
    return grp
