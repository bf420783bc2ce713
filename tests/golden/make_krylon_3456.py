"""Generates tests/golden/krylon_3456.npz from the CPU oracle: the one documented invocation of the
reference's built test program (README.md:223, src/tests/test_grasp_image.cpp:19-171) —
tutorials/krylon.pcd unvoxelised (tests/golden/krylon_xyz.npz), normals with radius 0.03 and then
negated, sample index 3456, one orientation, hand axes 0 1 2, 15 channels (SURVEY.md §8c names this
as the closest thing to a fixture the reference has).  The reference cannot be built here, so the
values are the oracle's ("parity unpinned"): the file pins the oracle against regressions and gives
the GPU test a committed target.

Run:  python tests/golden/make_krylon_3456.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle  # noqa: E402

xyz = np.load(os.path.join(HERE, "krylon_xyz.npz"))["xyz"]
p = oracle.default_params(15)
p.num_orientations = 1
p.num_hand_axes = 3
for i in range(3):
    p.hand_axes[i] = i
normals = -oracle.estimate_normals(xyz, radius=0.03)
hands = oracle.search(p, xyz, normals, np.array([3456], np.int32))
img, cand = oracle.images(p, xyz, normals, np.ones((1, len(xyz)), np.int32), np.zeros((1, 3)), hands.copy())
try:
    rev = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"]).decode().strip()
except Exception:
    rev = "unknown"
np.savez_compressed(os.path.join(HERE, "krylon_3456.npz"), normal_3456=normals[3456], normals_checksum=np.float64(normals.astype(np.float64).sum()),
                    hands=hands.view(np.uint8), images=img, cand_index=cand, oracle_git=np.array(rev))
print(hands["valid"], hands["finger_placement_index"], img.shape, cand)
