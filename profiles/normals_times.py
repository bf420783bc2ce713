"""gpd_hip_estimate_normals on the bench's 30k-point cloud (and on a raw 120k-point scan), a few calls: run under
`rocprofv3 --kernel-trace --stats` for the per-kernel times of VERDICT r3 item 3 (profiles/r04_normals_kernels.txt)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np

from gpd_amd import api, synth

ctx = api.Context(api.default_params(15))
for seed, n in ((4321, 30000), (4321, 120000)):
    cl = synth.make_cloud(seed, n)
    ctx.upload_cloud(cl["xyz"], np.zeros_like(cl["xyz"]), cl["cam_source"], cl["view_points"])
    ctx.estimate_normals(0.03)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.estimate_normals(0.03)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("points %d: wall incl. download %.3f ms (min of 5)" % (n, min(ts)))
ctx.close()
