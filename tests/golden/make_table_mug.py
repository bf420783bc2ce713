"""tests/golden/table_mug_xyz.npz: the reference's tutorials/table_mug.pcd (104 444 points, ASCII x y z) as float32 —
the un-voxelised scan used as the density stressor of tests/test_gpu_dense_scan.py (the reference tree is not on the
GPU box).  Run in the build container: python tests/golden/make_table_mug.py"""
import os

import numpy as np

SRC = "/root/reference/tutorials/table_mug.pcd"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "table_mug_xyz.npz")

with open(SRC) as f:
    for line in f:
        if line.startswith("DATA"):
            break
    xyz = np.loadtxt(f, dtype=np.float32, usecols=(0, 1, 2))
xyz = xyz[~np.isnan(xyz).any(1)]
assert xyz.shape == (104444, 3)
np.savez_compressed(OUT, xyz=xyz)
print(OUT, os.path.getsize(OUT))
