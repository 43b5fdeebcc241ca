"""TEST INFRASTRUCTURE ONLY -- records golden vectors of the hloc wire-format post-processing by EXECUTING the
reference's own functions.  `hloc/match_dense.py` cannot be imported here (h5py / torchvision / cv2 are absent), so the
pure-numpy helpers are pulled out of the reference source with `ast` at generation time and run in a scratch namespace;
nothing of the reference is copied into the repository.  Needs /root/reference (authoring container only).

    python oracle/make_golden_hloc.py   ->  tests/golden/hloc_formats.npz

Scenario: three images (A, B, C), three pairs of dense matches (AB, AC, BC) aggregated exactly like
`aggregate_matches` (match_dense.py:298-390) with max_error = 2, cell_size = 8: `assign_keypoints(update=True)` bins the
keypoints of every image, `kpids_to_matches0` builds matches0 / matching_scores0 per pair, the final keypoints of an image are
the most common bin entries; then `assign_matches`' nearest-neighbour re-assignment (update=False) on the final keypoints.
"""
import ast
import os
import sys
from collections import Counter, defaultdict
from typing import Dict, Iterable, List, Optional, Set, Tuple, Union

import numpy as np
from scipy.spatial import KDTree

REF = "/root/reference/hloc/match_dense.py"
WANT = {"to_cpts", "assign_keypoints", "get_grouped_ids", "get_unique_matches", "matches_to_matches0", "kpids_to_matches0"}
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hloc_formats.npz")


def reference_functions():
    tree = ast.parse(open(REF).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    assert {n.name for n in body} == WANT
    ns = {"np": np, "KDTree": KDTree, "Counter": Counter, "List": List, "Tuple": Tuple, "Union": Union, "Optional": Optional,
          "Dict": Dict, "Iterable": Iterable, "Set": Set}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return ns


def scenario(seed=0):
    """dense matches of the three pairs: [n,2] pixel keypoints per side (clustered, so bins collide) and scores"""
    g = np.random.default_rng(seed)
    sizes = {"A": (640, 480), "B": (512, 384), "C": (600, 450)}
    pairs = [("A", "B"), ("A", "C"), ("B", "C")]
    data = {}
    for k, (n0, n1) in enumerate(pairs):
        n = 400 + 50 * k
        centres0 = g.uniform(20, np.array(sizes[n0]) - 20, size=(60, 2))
        centres1 = g.uniform(20, np.array(sizes[n1]) - 20, size=(60, 2))
        idx = g.integers(0, 60, n)
        kp0 = (centres0[idx] + g.normal(0, 3.0, (n, 2))).astype(np.float32)
        kp1 = (centres1[idx] + g.normal(0, 3.0, (n, 2))).astype(np.float32)
        sc = g.uniform(0.05, 1.0, n).astype(np.float32)
        data[(n0, n1)] = (kp0, kp1, sc)
    return pairs, data


def main():
    R = reference_functions()
    conf = {"max_error": 2, "cell_size": 8}
    pairs, data = scenario()
    cpdict, bindict = defaultdict(list), defaultdict(list)
    pairs_per_q = Counter(n for p in pairs for n in p)
    out = {}
    for n0, n1 in pairs:
        kp0, kp1, sc = data[(n0, n1)]
        ids0 = R["assign_keypoints"](kp0, cpdict[n0], conf["max_error"], True, bindict[n0], sc, conf["cell_size"])
        ids1 = R["assign_keypoints"](kp1, cpdict[n1], conf["max_error"], True, bindict[n1], sc, conf["cell_size"])
        m0, s0 = R["kpids_to_matches0"](ids0, ids1, sc)
        tag = n0 + n1
        out.update({f"{tag}_kp0": kp0, f"{tag}_kp1": kp1, f"{tag}_scores": sc, f"{tag}_ids0": ids0, f"{tag}_ids1": ids1,
                    f"{tag}_matches0": m0, f"{tag}_scores0": s0})
        for name in (n0, n1):
            pairs_per_q[name] -= 1
            if pairs_per_q[name] > 0:
                continue
            kp_score = [c.most_common(1)[0][1] for c in bindict[name]]
            final = np.array([c.most_common(1)[0][0] for c in bindict[name]], dtype=np.float32)
            out[f"{name}_keypoints"], out[f"{name}_kp_score"] = final, np.array(kp_score, dtype=np.float64)
            cpdict[name] = final
    # assign_matches: nearest-neighbour re-assignment on the final keypoints
    for n0, n1 in pairs:
        kp0, kp1, sc = data[(n0, n1)]
        ids0 = R["assign_keypoints"](kp0, out[f"{n0}_keypoints"], conf["max_error"])
        ids1 = R["assign_keypoints"](kp1, out[f"{n1}_keypoints"], conf["max_error"])
        m0, s0 = R["kpids_to_matches0"](ids0, ids1, sc)
        out.update({f"{n0}{n1}_nn_ids0": ids0, f"{n0}{n1}_nn_matches0": m0, f"{n0}{n1}_nn_scores0": s0})
    # to_cpts on its own, incl. ps = 0
    pts = data[pairs[0]][0][:50]
    out["cpts8"], out["cpts2"], out["cpts0"] = (np.array(R["to_cpts"](pts, p), dtype=np.float64) for p in (8, 2, 0.0))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays;", {k: int((out[k] >= 0).sum()) for k in out if k.endswith("_matches0")})


if __name__ == "__main__":
    sys.exit(main())
