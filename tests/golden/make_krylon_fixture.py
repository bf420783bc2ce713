"""Packs the reference's tutorial cloud (tutorials/krylon.pcd, 4467 ASCII points x y z rgb) into
tests/golden/krylon_xyz.npz — the input of BASELINE.json configs[0].  Data fixture only; run in
the build container:  python tests/golden/make_krylon_fixture.py"""
import os
import sys

import numpy as np

SRC = "/root/reference/tutorials/krylon.pcd"
if not os.path.exists(SRC):
    sys.exit("reference not present: " + SRC)
rows = []
data = False
for line in open(SRC):
    if data:
        p = line.split()
        if len(p) >= 3:
            rows.append([float(p[0]), float(p[1]), float(p[2])])
    elif line.startswith("DATA"):
        assert "ascii" in line
        data = True
xyz = np.array(rows, np.float32)
assert xyz.shape == (4467, 3), xyz.shape
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "krylon_xyz.npz"), xyz=xyz)
print(xyz.shape, xyz.min(0), xyz.max(0))
