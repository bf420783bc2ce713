"""Generates tests/golden/case_small_c{15,12,3,1}.npz from the CPU oracle (the reference ships
no golden vectors and cannot be built here, SURVEY.md §4/§8c — "parity unpinned").
The fixture pins the oracle against regressions and gives the GPU tests a committed target.

Run:  python tests/golden/make_golden_case.py [channels ...]   (default: 15 12 3 1)
Inputs: synth.make_cloud(seed=4242, num_points=8000), first 10 sample indices, real LeNet
parameters (tests/golden/lenet{15,3}_params.npz) + synthetic ip1 (seed 42).
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from gpd_amd import synth  # noqa: E402

cl = synth.make_cloud(4242, 8000)
si = synth.sample_indices(cl, 10)
try:
    rev = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"]).decode().strip()
except Exception:
    rev = "unknown"
for C in ([int(a) for a in sys.argv[1:]] or (15, 12, 3, 1)):
    p = oracle.default_params(C)
    gold = os.path.join(HERE, "lenet%d_params.npz" % C)
    w = synth.lenet_weights(C, real=dict(np.load(gold)) if os.path.exists(gold) else None)
    hands = oracle.search(p, cl["xyz"], cl["normals"], si)
    hands_f = oracle.filter_workspace(p, hands.copy())
    img, cand = oracle.images(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], hands_f)
    scores = oracle.lenet(img, w)
    np.savez_compressed(os.path.join(HERE, "case_small_c%d.npz" % C), sample_indices=si, hands=hands.view(np.uint8),
                        hands_filtered_valid=hands_f["valid"], images=img, cand_index=cand, scores=scores,
                        oracle_git=np.array(rev))
    print(C, hands["valid"].sum(), img.shape, scores[:4])
